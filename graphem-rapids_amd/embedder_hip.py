"""GraphEmbedderHIP: host-side mirror of the reference's GraphEmbedderPyTorch
(graphem_rapids/backends/embedder_pytorch.py, "pt.py") whose layout loop runs in
hand-written HIP kernels on MI355X through the C ABI of include/graphem_hip.h.

Same constructor arguments, attributes, methods and exception types as pt.py:51-67,
776-844.  The loop body never touches PyTorch: torch is used for the
device handle, the zero-copy tensor view of the positions and the parity sampler.
"""
import logging

import numpy as np
import scipy.sparse as sp
import torch

from . import _native

logger = logging.getLogger(__name__)


class _DeviceArray:
    """Zero-copy description of engine memory for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner  # keeps the engine alive while a view exists


def device_view(ptr, shape, dtype, device, owner):
    typestr = {torch.float32: "<f4", torch.float64: "<f8", torch.int32: "<i4", torch.int64: "<i8",
               torch.uint8: "|u1"}[dtype]
    return torch.as_tensor(_DeviceArray(ptr, shape, typestr, owner), device=device)


class GraphEmbedderHIP:
    """Force-directed graph embedder; drop-in for GraphEmbedderPyTorch (pt.py:27-49)."""

    def __init__(
        self,
        adjacency,
        n_components=2,
        device=None,
        dtype=torch.float32,
        L_min=1.0,
        k_attr=0.2,
        k_inter=0.5,
        n_neighbors=10,
        sample_size=256,
        batch_size=None,
        memory_efficient=True,
        verbose=True,
        logger_instance=None,
        seed=None,
        *,
        sampler="auto",
        init="auto",
        knn_method="auto",
        knn_distance="auto",
        ivf_lists=0,
        ivf_probes=0,
    ):
        """Arguments as pt.py:51-104.  Extra keyword-only arguments:

        sampler : 'torch' draws each iteration's sample with torch.randperm(E)[:S] from the
            global CPU generator exactly as the reference's CPU backend does (pt.py:409);
            'device' uses the engine's on-GPU sampler (no host work in the loop);
            'auto' = 'torch' up to 2**20 edges, 'device' above.
        knn_method : 'scan' (exact filtered brute-force scan fused with the spring phase), 'grid' (n_components <= 3:
            exact search through a grid over the midpoints rebuilt every iteration), 'ivf' (n_components <= 16: an
            inverted-file index rebuilt every iteration, the counterpart of the reference's cuVS IVF-Flat,
            embedder_cuvs.py:255-313 -- `ivf_lists` centroids, every midpoint filed under the nearest; with
            `ivf_probes` > 0 (or 0 = engine default) a query sees that many lists and gets the exact k + 1 nearest among
            their members: APPROXIMATE; with `ivf_probes` < 0 it sees every list that can hold a neighbour: EXACT, the
            rows of 'scan'), or 'auto' = exact methods only: the exact inverted file for 2-8 components, >= 262144 edges
            and thousands of sampled midpoints (sample_size >= 4096 up to 4 components, >= 8192 for 5-8), else 'grid' when n_components <= 3 and sample_size >= 12288, else 'scan'.  The sub-quadratic searches pay
            from a few thousand sampled midpoints on.  'ivf' is not available with knn_distance='cdist'.
        knn_distance : 'cdist' ranks the neighbours on the value torch.cdist gives (ATen's matmul form, fp32) and orders
            equal values as torch.topk does, i.e. the neighbour ids of the reference's PyTorch-CPU backend row for row
            (pt.py:580-583); 'exact' ranks on the exact-difference squared distance, ties on the smaller id (what the
            reference's KeOps path computes, pt.py:527-534; one launch less per iteration).  'auto' = 'cdist' with
            sampler='torch' (the parity mode), 'exact' with sampler='device' (the speed mode).
        init : 'laplacian' (scipy eigsh exactly as pt.py:337-379), 'laplacian_hip' (the same
            eigenvectors by thick-restart Lanczos on the GPU, spectral.py: 1.2 s at 100 K vertices where
            eigsh takes 29 s, 3.4 s at 1 M where it is impractical), 'random' (the reference's own
            fallback, pt.py:369), or 'auto' = 'laplacian' up to 20000 vertices, 'laplacian_hip' above.
        """
        if seed is not None:  # pt.py:106-111
            np.random.seed(seed)
            torch.manual_seed(seed)
        self.seed = seed

        # device plumbing (pt.py:113-117): the HIP engine has no CPU path
        dev = torch.device("cuda" if device is None else device)  # invalid strings raise RuntimeError here
        if dev.type != "cuda":
            raise RuntimeError(f"GraphEmbedderHIP needs a GPU device, got '{dev}' (no CPU fallback)")
        self.device = torch.device("cuda", dev.index if dev.index is not None else 0)

        if logger_instance is not None:
            self.logger = logger_instance
        else:
            self.logger = logger
            if verbose:
                logging.basicConfig(level=logging.INFO)

        adjacency = self._validate_adjacency(adjacency)
        self.adjacency = adjacency
        self.n = adjacency.shape[0]
        self.n_components = n_components
        if dtype not in (torch.float32, torch.float64, torch.float16):
            raise ValueError(f"unsupported dtype {dtype}")
        # pt.py:56: the reference computes in the dtype it is given.  float32: the fused engine; float64: the float64
        # engine (csrc/f64.hip, every phase in double); float16: computed in float32, returned in float16 (logged)
        self.dtype = dtype
        if dtype == torch.float16:
            self._dtype_note = "GraphEmbedderHIP computes in float32; dtype=torch.float16 only sets the dtype of the positions it returns"
            logging.getLogger(__name__).warning(self._dtype_note)
        self.L_min = L_min
        self.k_attr = k_attr
        self.k_inter = k_inter
        self.n_neighbors = n_neighbors
        self.memory_efficient = memory_efficient
        self.batch_size = batch_size

        if n_components <= 0:  # pt.py:143-146
            raise ValueError(f"Number of components must be positive, got {n_components}")
        if k_attr < 0:
            raise ValueError(f"Attractive force constant k_attr must be non-negative, got {k_attr}")
        self.verbose = verbose

        edges = self._extract_edges_from_adjacency(adjacency)
        self.n_edges = len(edges)
        self.sample_size = min(sample_size, self.n_edges)  # pt.py:156
        self._edges_np = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
        self.edges = torch.from_numpy(self._edges_np.astype(np.int64))  # (E, 2) integer tensor like pt.py:159
        self._has_pykeops = False  # attribute read by callers (pt.py:162); KeOps is never used here
        if self.batch_size is None:
            self.batch_size = self.n  # the engine never chunks the query set

        if knn_method not in _native.KNN_METHOD:
            raise ValueError(f"Invalid knn_method: {knn_method}")
        self.knn_method = knn_method
        if sampler not in ("auto", "torch", "device"):
            raise ValueError(f"Invalid sampler: {sampler}")
        if sampler == "auto":
            sampler = "torch" if self.n_edges <= (1 << 20) else "device"
        self.sampler = sampler
        if knn_distance not in ("auto", "exact", "cdist"):
            raise ValueError(f"Invalid knn_distance: {knn_distance}")
        # knn_distance='cdist' (the parity mode) ranks the candidates of the fused brute-force scan: it cannot be combined with
        # another search.  An explicit 'grid' / 'ivf' therefore excludes it (ValueError when both are explicit), and 'auto'
        # resolves to 'cdist' only where knn_method='auto' / 'scan' keeps the engine on the scan anyway -- the same size
        # rule as GH_KNN_AUTO in csrc/api.hip (thousands of sampled midpoints go to the exact inverted file / the grid).
        big_s_ivf = (2 <= n_components <= 8 and self.sample_size >= (4096 if n_components <= 4 else 8192)
                     and self.n_edges >= 262144)
        big_s_grid = n_components <= 3 and self.sample_size >= 12288
        stays_on_scan = knn_method == "scan" or (knn_method == "auto" and not (big_s_ivf or big_s_grid))
        if knn_distance == "cdist" and knn_method in ("grid", "ivf"):
            raise ValueError(f"knn_method='{knn_method}' cannot be combined with knn_distance='cdist' (the parity mode "
                             "re-values the candidates of the brute-force scan); use knn_method='scan' or 'auto'")
        if knn_distance == "auto":
            knn_distance = "cdist" if sampler == "torch" and stays_on_scan else "exact"
        elif knn_distance == "cdist" and not stays_on_scan:
            self.logger.warning("knn_distance='cdist' keeps the KNN on the brute-force scan: with sample_size=%d the "
                                "sub-quadratic search knn_method='auto' would choose is not used", self.sample_size)
        self.knn_distance = knn_distance

        # the device sampler's key: an unseeded embedder draws it from torch's global generator, so unseeded
        # runs differ from each other (like the reference's) while torch.manual_seed still controls both samplers
        # (drawn only for the device sampler: with sampler='torch' the draw would shift the randperm stream the reference's
        # CPU backend consumes after torch.manual_seed, pt.py:409)
        if seed is not None:
            self._engine_seed = int(seed)
        elif self.sampler == "device":
            self._engine_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        else:
            self._engine_seed = 0
        self._engine = _native.Engine(
            self.n, n_components, self._edges_np, L_min, k_attr, k_inter, n_neighbors, self.sample_size,
            seed=self._engine_seed, device_id=self.device.index, knn_method=knn_method,
            knn_distance="exact" if dtype == torch.float64 else knn_distance, ivf_lists=ivf_lists, ivf_probes=ivf_probes,
            dtype="float64" if dtype == torch.float64 else "float32")
        if self.verbose:
            self.logger.info("Initialized GraphEmbedderHIP on %s", self.device)
            self.logger.info("Graph: %d vertices, %d edges, %dD", self.n, self.n_edges, self.n_components)

        if init == "auto":
            init = "laplacian" if self.n <= 20000 else "laplacian_hip"
        if init == "laplacian":
            p0 = self._compute_laplacian_embedding()
        elif init == "laplacian_hip":
            from .spectral import laplacian_embedding_hip
            try:
                p0 = laplacian_embedding_hip(self.adjacency, self.n_components, device=str(self.device),
                                             seed=0 if seed is None else seed, dtype=np.float64 if dtype == torch.float64 else np.float32)
            except ValueError as exc:  # n_components + 1 >= n: the reference falls back to random too
                self.logger.warning("Eigendecomposition failed: %s", exc)
                p0 = np.random.randn(self.n, self.n_components) * 0.1
        elif init == "random":
            p0 = np.random.randn(self.n, self.n_components) * 0.1   # (float64, like pt.py:369; the engine casts to its dtype)
        else:
            raise ValueError(f"Invalid init: {init}")
        self._engine.set_positions(p0)   # (a float64 engine takes the start as float64: pt.py:372-376 casts it to dtype)

    # ---- construction helpers (boundary, host side) -----------------------------------
    @staticmethod
    def _validate_adjacency(adjacency):
        """Any sparse / dense / array-like square matrix -> CSR (contract of pt.py:182-218)."""
        if sp.issparse(adjacency):
            adjacency = adjacency.tocsr()
        elif not isinstance(adjacency, np.ndarray):
            adjacency = np.asarray(adjacency)
        if adjacency.ndim != 2 or adjacency.shape[0] != adjacency.shape[1]:
            raise ValueError(f"Adjacency matrix must be square, got shape {adjacency.shape}")
        if adjacency.shape[0] == 0:
            raise ValueError("Adjacency matrix cannot be empty")
        if not sp.issparse(adjacency):
            adjacency = sp.csr_matrix(adjacency)
        return adjacency

    def _extract_edges_from_adjacency(self, adjacency):
        """Upper triangle of the nonzero pattern as given, in CSR row order (pt.py:220-245)."""
        rows, cols = adjacency.nonzero()
        keep = rows < cols
        edges = np.column_stack([rows[keep], cols[keep]])
        if self.verbose and len(edges) == 0:
            self.logger.warning("No edges found in adjacency matrix")
        return edges

    def _compute_laplacian_embedding(self):
        """Spectral start: eigenvectors 1..D of the normalised Laplacian of the symmetrised,
        unweighted graph; random fallback if the eigensolver fails (contract of pt.py:337-379)."""
        import scipy.sparse.linalg as spla
        from scipy.sparse.csgraph import laplacian
        sym = sp.csr_matrix(self.adjacency + self.adjacency.transpose())
        sym.data = np.ones_like(sym.data)
        lap = laplacian(sym, normed=True)
        want = self.n_components + 1
        try:
            _, vecs = spla.eigsh(lap, want, which="SM")
            emb = vecs[:, 1:want]
        except Exception as exc:  # pylint: disable=broad-exception-caught
            self.logger.warning("Eigendecomposition failed: %s", exc)
            emb = np.random.randn(self.n, self.n_components) * 0.1
        # pt.py:372-376 casts the float64 eigenvectors straight to dtype: a float64 engine gets them unrounded
        return np.ascontiguousarray(emb, dtype=np.float64 if self.dtype == torch.float64 else np.float32)

    # ---- positions accessors (pt.py:324-335, 835-844) -----------------------------------
    @property
    def positions(self):
        """Host numpy copy of the positions, shape (n, D)."""
        out = self._engine.get_positions()   # float64 from a float64 engine
        return out.astype(np.float16) if self.dtype == torch.float16 else out

    @positions.setter
    def positions(self, value):
        if isinstance(value, torch.Tensor):
            value = value.detach().to("cpu", torch.float64 if self._engine.f64 else torch.float32).numpy()
        self._engine.set_positions(np.asarray(value, dtype=self._engine.np_dtype))

    @property
    def _positions(self):
        """Device tensor of the positions (callers and tests read .device / .dtype / values)."""
        if self._engine.f64:
            self._engine.sync()
            return device_view(self._engine.positions_device_ptr(), (self.n, self.n_components), torch.float64,
                               self.device, self._engine).clone()
        view = device_view(self._engine.positions_unpadded_device_ptr(), (self.n, self.n_components), torch.float32,
                           self.device, self._engine).clone()  # the engine reuses that buffer
        return view if self.dtype == torch.float32 else view.to(self.dtype)

    def get_positions(self):
        return self.positions

    # ---- the loop (pt.py:776-833) -------------------------------------------------------
    def _torch_rng_state(self):
        """The global CPU generator's state as a writable uint8 array when it is the 5056-byte mt19937 blob
        gh_torch_randperm_prefix understands, else None (the caller then asks torch.randperm itself)."""
        st = torch.get_rng_state()
        if st.dtype != torch.uint8 or st.numel() != 5056:
            return None
        return st.numpy().copy()

    def _draw_samples(self, iterations):
        """Sample ids for `iterations` iterations, consuming the global torch CPU generator
        exactly as pt.py:409 does (one randperm(E) per iteration; none when S >= E).  The ids are
        torch.randperm(E)[:S]; only the S entries asked for are materialised (ATen's CPU randperm is a forward
        Fisher-Yates: gh_torch_randperm_prefix, include/graphem_hip.h), and the generator is left in the state
        `iterations` calls of torch.randperm(E) leave it in."""
        E, S = self.n_edges, self.sample_size
        if S >= E or self.sampler == "device":
            return None
        state = self._torch_rng_state()
        if state is not None:
            try:
                out = _native.torch_randperm_prefix(state, E, S, iterations)
            except ValueError:
                state = None
            else:
                torch.set_rng_state(torch.from_numpy(state))
                return out
        out = np.empty((iterations, S), dtype=np.int32)   # a generator this build does not know: ask torch (ms per draw)
        for t in range(iterations):
            out[t] = torch.randperm(E)[:S].numpy()
        return out

    def update_positions(self):
        """One layout iteration (pt.py:776-806)."""
        ids = self._draw_samples(1)
        self._engine.step(None if ids is None else ids[0])

    def run_layout(self, num_iterations=100):
        """num_iterations iterations without host synchronisation; returns (n, D) numpy (pt.py:808-833).
        sampler='torch': the ids are drawn by a host thread of the library while the GPU runs the iterations before
        (gh_run_torch_sampled); the global CPU generator ends where the reference's loop would leave it."""
        if self.verbose:
            self.logger.info("Running layout for %d iterations", num_iterations)
        if num_iterations > 0:
            state = self._torch_rng_state() if (self.sampler == "torch" and self.sample_size < self.n_edges) else None
            if state is not None:
                try:
                    self._engine.run_torch_sampled(num_iterations, state)
                except ValueError as exc:
                    if "rng_state" not in str(exc):
                        raise
                    state = None
                else:
                    torch.set_rng_state(torch.from_numpy(state))
            if state is None:
                self._engine.run(num_iterations, self._draw_samples(num_iterations))
        return self.positions

    # ---- per-phase access (used by the parity tests; same names as pt.py) ------------------
    def _compute_spring_forces(self):
        return self._engine.spring_forces()

    def _locate_knn_midpoints(self, sampled_indices=None):
        """(knn_indices (S, k), sampled_indices (S,)) like pt.py:381-424."""
        if sampled_indices is None:
            ids = self._draw_samples(1)
            sampled_indices = np.arange(self.n_edges, dtype=np.int32) if ids is None else ids[0]
        sampled_indices = np.asarray(sampled_indices, dtype=np.int32)
        return self._engine.knn_midpoints(sampled_indices), sampled_indices

    def _compute_intersection_forces(self, knn_indices, sampled_indices):
        return self._engine.intersection_forces(sampled_indices, knn_indices)

    # ---- point-set KNN helpers callers and tests reach for (pt.py:260-322, 426-483, 543-593) --------
    def _get_adaptive_chunk_size(self, n_query, n_ref, backend):
        """The engine never chunks the query set; kept for interface parity (always > 0)."""
        return max(1, min(int(self.batch_size), int(n_query)) if self.batch_size else int(n_query))

    def _compute_knn_torch(self, query_points, reference_points, k, chunk_size=None):
        """(n_query, k) long tensor of nearest reference rows, ascending distance (pt.py:543-593)."""
        q = query_points.detach().to("cpu", torch.float32).numpy() if isinstance(query_points, torch.Tensor) \
            else np.asarray(query_points, dtype=np.float32)
        r = reference_points.detach().to("cpu", torch.float32).numpy() if isinstance(reference_points, torch.Tensor) \
            else np.asarray(reference_points, dtype=np.float32)
        idx = _native.knn_points(q, r, k, device_id=self.device.index)
        return torch.from_numpy(idx).to(self.device)

    def _compute_knn_pykeops(self, query_points, reference_points, k, chunk_size=None):
        """(n_query, k) nearest reference rows by the exact-difference squared distance, ties on the smaller index -- what
        the reference's KeOps path computes (`((x_i - y_j) ** 2).sum(-1).argKmin(k)`, pt.py:485-541); here it is the same
        kernel as _compute_knn_torch (gh_knn_points ranks on exactly that distance), KeOps itself is never imported."""
        return self._compute_knn_torch(query_points, reference_points, k, chunk_size)

    def _compute_knn_chunked(self, query_points, reference_points, k):
        """Same result as _compute_knn_torch: there is one KNN implementation here (pt.py:426-483)."""
        return self._compute_knn_torch(query_points, reference_points, k)

    def kernel_timings(self):
        return self._engine.timings()

    def display_layout(self, edge_width=1, node_size=3, node_colors=None):
        """Plotly rendering is outside the accelerated path."""
        if self.n_components not in (2, 3):
            raise ValueError("Display only supports 2D or 3D embeddings")  # pt.py:846-871
        raise NotImplementedError("display_layout is not part of the HIP backend; plot get_positions() instead")

    def __repr__(self):
        return (f"GraphEmbedderHIP(n_vertices={self.n}, n_components={self.n_components}, "
                f"n_edges={self.n_edges}, device={self.device})")
