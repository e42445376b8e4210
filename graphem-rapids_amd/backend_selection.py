"""Backend names and selection for the HIP drop-in (counterpart of the reference's
graphem_rapids/utils/backend_selection.py:16-29, 142-206; heuristics are out of scope,
SURVEY.md section 2 row 5: this package ships exactly one backend, 'hip')."""
from dataclasses import dataclass

from . import _native

VALID_BACKENDS = ["pytorch", "cuvs", "cpu", "auto", "hip"]


@dataclass
class BackendConfig:
    """Same fields as the reference's BackendConfig, with 'hip' accepted as a backend name."""
    n_vertices: int
    n_components: int = 2
    force_backend: str = None
    prefer_gpu: bool = True
    memory_limit: float = None  # GB
    verbose: bool = True

    def __post_init__(self):
        if self.force_backend and self.force_backend not in VALID_BACKENDS:
            raise ValueError(f"Invalid backend: {self.force_backend}")


def check_hip_availability():
    """Is the native library built and is a GPU visible?  Never raises."""
    info = {"library": False, "device_count": 0, "version": None}
    try:
        lib = _native.load()
        info["library"] = True
        info["version"] = lib.gh_version().decode()
        import torch
        info["device_count"] = torch.cuda.device_count()  # does not initialise the GPU
    except Exception:  # pylint: disable=broad-exception-caught
        pass
    return info


def get_optimal_backend(config):
    """'hip' whenever it is asked for or left to choose; the reference's own backends are not
    part of this package."""
    forced = config.force_backend
    if forced in (None, "auto", "hip"):
        return "hip"
    if forced not in VALID_BACKENDS:
        raise ValueError(f"Invalid backend: {forced}")
    raise ValueError(f"backend '{forced}' belongs to graphem_rapids itself; graphem_rapids_amd provides 'hip'")


def estimate_memory_usage(n_vertices, n_components, n_edges=None, n_neighbors=10, sample_size=256, knn_method="auto",
                          knn_distance="exact"):
    """Bytes of HBM the engine allocates (DESIGN.md 'Data layout'), including what grows with sample_size: the candidate
    lists of the filtered scan (128 KiB per sampled midpoint: 2 GiB at 16384), the buffers of knn_distance='cdist' (values of
    the replayed rows' prefixes: up to min(sample_size, 2^30 / E) rows of E floats) and the inverted file of knn_method='ivf'
    (or 'auto' with thousands of sampled midpoints)."""
    ld = 4 if n_components <= 4 else 8 if n_components <= 8 else 16 if n_components <= 16 else (n_components + 3) // 4 * 4
    e = n_edges if n_edges is not None else 5 * n_vertices
    s = min(sample_size, e)
    k = n_neighbors
    per_vertex = ld * 4 * 4 + n_components * 4 + ld * 8 + 4   # pos, new, 2 scratch, io, fp64 accumulators, flag
    scan = e >= 16384 and n_components >= 2 and ld <= 16 and k + 1 + (1 if knn_distance == "cdist" else 0) <= 128   # csrc/knn.hip gh_knn_scan_path
    per_edge = 8 + 8 + 4 * ld + (2 * ld if scan else 0)       # edge list, pull lists, midpoints, threshold subset
    per_query = (16384 * 8 if scan else 0) + 32 * 4 + (k + 1) * 16 + k * (4 + 4 * ld) + 16 * k   # candidate list (GH_CAND_CAP keys), counters, keys, pair scratch, touched list
    if s >= 2048:
        per_query += 8 + 16 * max(k, 1)                        # per-query runs of the touched list
    total = n_vertices * per_vertex + e * per_edge + s * per_query
    if knn_distance == "cdist":
        vstride = (e + 1023) // 1024 * 1024
        rows = min(s, max(16, (1 << 30) // max(vstride, 1)))
        total += rows * vstride * 4 + rows * (vstride // 64) * 4 + 12 * s
    ivf = knn_method == "ivf" or (knn_method == "auto" and knn_distance == "exact" and 2 <= n_components <= 8
                                  and s >= (4096 if n_components <= 4 else 8192) and e >= 262144)
    if ivf:
        lists = 512 if n_components <= 4 else 1024
        cap_rows = e + 512 * lists
        total += 8 * e + cap_rows * (4 * ld + 4) + 3 * 4 * s * lists   # assignment, members in list order, (query, list) pairs of the exact mode
    return total
