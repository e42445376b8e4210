"""Thin counterparts of the reference's memory utilities (graphem_rapids/utils/memory_management.py).

In the reference these wrap every hot function and are where the CPU time goes: MemoryManager.__exit__
calls gc.collect() four times per iteration (98.6 % of the wall time at n = 1000).
The HIP engine allocates everything once in gh_create and never touches the allocator or the garbage
collector inside the loop, so these are kept only so that caller code written against the reference
keeps working; none of them is called by the layout loop.
"""
import functools
import logging

import torch

logger = logging.getLogger(__name__)


def get_gpu_memory_info():
    """Same keys as the reference (memory_management.py:14-42): GB figures of device 0."""
    info = {"available": False, "total": 0.0, "allocated": 0.0, "cached": 0.0, "free": 0.0}
    try:
        if torch.cuda.is_available():
            free, total = torch.cuda.mem_get_info()
            info.update(available=True, total=total / 1024 ** 3, free=free / 1024 ** 3,
                        allocated=torch.cuda.memory_allocated() / 1024 ** 3,
                        cached=torch.cuda.memory_reserved() / 1024 ** 3)
    except Exception:  # pylint: disable=broad-exception-caught
        pass
    return info


def get_optimal_chunk_size(n_vertices, n_components, available_memory_gb=None, safety_factor=0.7, backend="hip"):
    """The engine never chunks its query set (the (S, E) distance matrix is never materialised), so the
    whole vertex count is the 'chunk'; signature of memory_management.py:45-114."""
    del n_components, available_memory_gb, safety_factor, backend
    return max(1, int(n_vertices))


def cleanup_gpu_memory():
    """Releases torch's cached blocks; no gc.collect() (memory_management.py:117-128 calls it every time)."""
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def monitor_memory_usage(func):
    """Decorator with the reference's name (memory_management.py:131-167): logs the allocation delta at DEBUG
    level, adds no synchronisation."""
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        before = torch.cuda.memory_allocated() if torch.cuda.is_available() else 0
        out = func(*args, **kwargs)
        if torch.cuda.is_available():
            logger.debug("%s: %+.3f GB", func.__name__, (torch.cuda.memory_allocated() - before) / 1024 ** 3)
        return out
    return wrapper


class MemoryManager:
    """Context manager with the reference's interface (memory_management.py:170-208); leaving it never
    triggers a garbage collection or a device synchronisation."""

    def __init__(self, cleanup_on_exit=False):
        self.cleanup_on_exit = cleanup_on_exit
        self.initial_memory = None

    def __enter__(self):
        self.initial_memory = get_gpu_memory_info()
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.cleanup_on_exit:
            cleanup_gpu_memory()
        return False
