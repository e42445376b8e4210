"""graphem_rapids_amd: MI355X-native force-directed graph embedding.

Drop-in for one path of sashakolpakov/graphem-rapids: create_graphem() /
run_layout() / get_positions() (reference graphem_rapids/__init__.py:78-136), backed by
hand-written HIP kernels for gfx950 behind the C ABI in include/graphem_hip.h.
"""
from .backend_selection import BackendConfig, check_hip_availability, get_optimal_backend, estimate_memory_usage
from .embedder_hip import GraphEmbedderHIP
from .memory_management import (MemoryManager, cleanup_gpu_memory, get_gpu_memory_info, get_optimal_chunk_size,
                                monitor_memory_usage)
from .generators import (erdos_renyi_graph, generate_random_regular, erdos_renyi_edges, random_regular_edges, planted_partition_edges,
                         edges_to_adjacency, load_snap_edge_list)

__version__ = "0.1.0"


def create_graphem(adjacency, n_components=2, backend=None, **kwargs):
    """Same factory signature as the reference (graphem_rapids/__init__.py:78-136).
    backend: None / 'auto' / 'hip' -> GraphEmbedderHIP; anything else raises ValueError."""
    config = BackendConfig(n_vertices=adjacency.shape[0], n_components=n_components)
    config.force_backend = backend
    get_optimal_backend(config)  # validates the name
    return GraphEmbedderHIP(adjacency, n_components, **kwargs)


def get_backend_info():
    """Availability report, shaped like the reference's get_backend_info (__init__.py:139-169)."""
    hip = check_hip_availability()
    return {
        "hip_library": hip["library"],
        "hip_version": hip["version"],
        "cuda_available": hip["device_count"] > 0,
        "cuda_device_count": hip["device_count"],
        "recommended_backend": "hip" if hip["library"] and hip["device_count"] > 0 else None,
    }


def graphem_seed_selection(embedder, k, num_iterations=20):
    """Caller contract of the reference's influence.graphem_seed_selection (influence.py:10-37):
    run the layout, pick the k vertices farthest from the origin."""
    import numpy as np
    embedder.run_layout(num_iterations=num_iterations)
    engine = getattr(embedder, "_engine", None)
    if engine is not None and 1 <= k <= min(64, embedder.n):
        # radial norm + top-k on the device, k ids come back instead of (n, D) positions
        return engine.radial_topk(k).tolist()
    radial = np.linalg.norm(np.array(embedder.positions), axis=1)
    return np.argsort(-radial)[:k].tolist()


__all__ = ["create_graphem", "get_backend_info", "GraphEmbedderHIP", "BackendConfig", "get_optimal_backend",
           "check_hip_availability", "estimate_memory_usage", "erdos_renyi_graph", "generate_random_regular",
           "erdos_renyi_edges", "random_regular_edges", "planted_partition_edges", "edges_to_adjacency", "load_snap_edge_list",
           "graphem_seed_selection", "MemoryManager", "cleanup_gpu_memory", "get_gpu_memory_info",
           "get_optimal_chunk_size", "monitor_memory_usage"]
