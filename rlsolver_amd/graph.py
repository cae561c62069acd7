"""Graph / instance IO and the CSR layouts the HIP kernels consume.

Host-side only (numpy); nothing here touches the GPU.  Mirrors the reference's
L0 data layer so callers can swap imports:

* ``read_mygraph``            <- rlsolver/methods/util_read_data.py:76-82
* ``load_mygraph2``           <- rlsolver/methods/util_read_data.py:121-140
* ``calc_num_nodes_in_mygraph`` <- rlsolver/methods/util.py:35-40
* ``build_adjacency_indies``  <- rlsolver/methods/util_read_data.py:144-187
* ``build_adjacency_bool``    <- rlsolver/methods/util.py:343-370
* ``read_tsp_file``/``load_tsp`` <- rlsolver/methods/ISCO/util_TSP.py:5-45

plus deterministic synthetic generators (the real Gset files are not in the
reference) and ``GraphCSR`` -- the device layout:

    edges "as stored":  eu[E'], ev[E'] int32      (objective kernels, K1/K8)
    symmetric CSR:      rowptr[N+1], col[2E] int32 (delta kernels, K2-K7)

Multi-edges are kept with multiplicity (the reference counts a duplicated edge
twice); self loops never change an XOR so they are dropped from the CSR but
kept out of nothing else (x_u ^ x_u == 0).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

MyGraph = List[Tuple[int, int, int]]  # rlsolver/methods/config.py:5


# --------------------------------------------------------------------------- #
# text formats
# --------------------------------------------------------------------------- #
def _graph_lines(filename: str):
    with open(filename, "r") as f:
        for raw in f:
            line = raw.split("//", 1)[0].strip()
            if line:
                yield line


def read_graph_header(filename: str) -> Tuple[int, int]:
    """(num_nodes, num_edges) from the first non-comment line of a Gset-style file."""
    for line in _graph_lines(filename):
        p = line.split()
        return int(p[0]), int(p[1])
    raise ValueError(f"empty graph file {filename}")


def read_mygraph(filename: str) -> MyGraph:
    """Gset text -> list of (n0, n1, weight), 0-based ids.

    Same result as the reference reader (util_read_data.py:76-82) on the files
    it accepts; additionally tolerates ``//`` comment lines the way
    ``read_nxgraph`` (util_read_data.py:46-66) does.
    """
    it = _graph_lines(filename)
    try:
        next(it)  # header
    except StopIteration:
        raise ValueError(f"empty graph file {filename}")
    mygraph: MyGraph = []
    for line in it:
        p = line.split()
        n0, n1 = int(p[0]), int(p[1])
        w = int(float(p[2])) if len(p) > 2 else 1
        mygraph.append((n0 - 1, n1 - 1, w))
    return mygraph


def read_edge_arrays(filename: str, chunk_bytes: int = 64 << 20) -> Tuple[int, np.ndarray, np.ndarray, np.ndarray]:
    """Streaming reader for 10^4 .. 10^6-node files: (num_nodes_from_header, eu, ev, w) as int32 arrays, 0-based.

    The file is consumed in blocks of ``chunk_bytes`` cut at a line end; a block's tokens are split in C
    (``bytes.split``) and converted by numpy in one call, so memory stays at one block + the output arrays and a
    10^7-edge file parses in seconds (the reference's readers build a Python tuple per edge, util_read_data.py:76-82).
    ``//`` comments and blank lines are tolerated; a block that contains a comment falls back to per-line filtering.
    Rows are ``n0 n1 [weight]`` (weight defaults to 1; float weights such as "1.0" are accepted and truncated like the
    reference's ``int(float(.))``)."""
    n = m = None
    us, vs, ws = [], [], []
    carry = b""
    with open(filename, "rb") as f:
        while True:
            block = f.read(chunk_bytes)
            if not block and not carry:
                break
            buf = carry + block
            if block:
                cut = buf.rfind(b"\n")
                if cut < 0:
                    carry = buf
                    continue
                buf, carry = buf[:cut + 1], buf[cut + 1:]
            else:
                carry = b""
            if b"/" in buf:      # comments present: filter this block line by line
                buf = b"\n".join(ln.split(b"//", 1)[0] for ln in buf.split(b"\n"))
            if n is None:        # header = first non-blank line
                stripped = buf.lstrip()
                if not stripped:
                    continue
                head, _, buf = stripped.partition(b"\n")
                hp = head.split()
                n, m = int(hp[0]), int(hp[1])
            tok = buf.split()
            if not tok:
                continue
            # fast path: every line of the block has the token count of its first data row (2, or 3 with a weight) --
            # true exactly when rows * cols tokens sit on `rows` lines.  Anything else (blank lines, rows with and
            # without a weight mixed) is parsed line by line like the reference's reader, which raises on a malformed row.
            cols = len(buf.lstrip().split(b"\n", 1)[0].split())
            lines = buf.count(b"\n") + (0 if buf.endswith(b"\n") else 1)
            if cols in (2, 3) and len(tok) == cols * lines:
                try:
                    a = np.array(tok, dtype=np.int64).reshape(-1, cols)
                except (ValueError, OverflowError):
                    a = np.array(tok, dtype=np.float64).reshape(-1, cols).astype(np.int64)
            else:
                rows = []
                for ln in buf.split(b"\n"):
                    p = ln.split()
                    if not p:
                        continue
                    if len(p) not in (2, 3):
                        raise ValueError(f"{filename}: rows must be 'n0 n1 [weight]', got {ln.decode(errors='replace')!r}")
                    rows.append((int(p[0]), int(p[1]), int(float(p[2])) if len(p) == 3 else 1))
                a = np.array(rows, dtype=np.int64).reshape(-1, 3)
                cols = 3
            if a.size and (a[:, :2].min() < 1 or a[:, :2].max() > n):
                bad = a[((a[:, :2] < 1) | (a[:, :2] > n)).any(axis=1)][0]
                raise ValueError(f"{filename}: node id outside [1, {n}] in row {bad[0]} {bad[1]}")
            us.append((a[:, 0] - 1).astype(np.int32))
            vs.append((a[:, 1] - 1).astype(np.int32))
            ws.append(a[:, 2].astype(np.int32) if cols == 3 else np.ones(len(a), np.int32))
    if n is None:
        raise ValueError(f"empty graph file {filename}")
    total = sum(len(u) for u in us)
    if m is not None and total != m:
        import warnings
        warnings.warn(f"{filename}: header announces {m} edges, the file holds {total}")
    if not us:
        z = np.zeros(0, np.int32)
        return n, z, z.copy(), z.copy()
    return n, np.concatenate(us), np.concatenate(vs), np.concatenate(ws)


def write_mygraph(filename: str, mygraph: Sequence[Tuple[int, int, int]], num_nodes: int) -> None:
    with open(filename, "w") as f:
        f.write(f"{num_nodes} {len(mygraph)}\n")
        for n0, n1, w in mygraph:
            f.write(f"{n0 + 1} {n1 + 1} {w}\n")


def calc_num_nodes_in_mygraph(mygraph: MyGraph) -> int:
    """Number of *distinct endpoints* -- NOT the header N (util.py:35-40).

    The reference sizes EnvMaxcut with this, so a graph with isolated nodes is
    mis-sized there; we reproduce the value and let callers pass the header N
    explicitly when they have it.
    """
    s = set()
    for n0, n1, _ in mygraph:
        s.add(n0)
        s.add(n1)
    return len(s)


def load_mygraph2(dataDir: str = "./data/syn_PL", graph_name: str = "") -> MyGraph:
    """File lookup + synthetic fallback, as util_read_data.py:121-140.

    ``<dataDir>/<graph_name>.txt`` or a literal ``*.txt`` path is read; names
    like ``BA_100_ID3`` are generated (numpy generators below, seeded by the ID
    -- the reference seeds ``random`` and calls networkx, so generated graphs
    are same-distribution, not same-edges).
    """
    p = f"{dataDir}/{graph_name}.txt"
    if os.path.exists(p):
        return read_mygraph(p)
    if os.path.isfile(graph_name) and os.path.splitext(graph_name)[-1] == ".txt":
        return read_mygraph(graph_name)
    parts = graph_name.split("_")
    if len(parts) >= 2 and parts[0] in ("BA", "ER", "PL"):
        seed = int(parts[-1][2:]) if parts[-1].startswith("ID") else None
        num_nodes = int(parts[-2] if seed is not None else parts[-1])
        return generate_mygraph(parts[0], num_nodes, seed=seed)[0]
    raise ValueError(f"DataDir {dataDir} | graph_name {graph_name} txt_path {p}")


def load_mygraph(DataDir: str, graph_name: str) -> MyGraph:
    """util_read_data.py:98-119: ``<DataDir>/<graph_name>.txt``; else a name ``<type>_<n>_ID<i>`` or ``<type>_<n>`` is
    generated (see load_mygraph2 on what "generated" means here); else ``graph_name`` itself as a path."""
    p = f"{DataDir}/{graph_name}.txt"
    if os.path.exists(p):
        return read_mygraph(p)
    parts = graph_name.split("_")
    if parts[0] in ("BA", "ER", "PL") and len(parts) == 3:
        return generate_mygraph(parts[0], int(parts[1]), seed=int(parts[2][len("ID"):]))[0]
    if parts[0] in ("BA", "ER", "PL") and len(parts) == 2:
        return generate_mygraph(parts[0], int(parts[1]))[0]
    if os.path.isfile(graph_name):
        return read_mygraph(graph_name)
    raise ValueError(f"DataDir {DataDir} | graph_name {graph_name}")


# --------------------------------------------------------------------------- #
# adjacency forms of the reference surface
# --------------------------------------------------------------------------- #
def build_adjacency_indies(mygraph: MyGraph, if_bidirectional: bool = False, num_nodes: int = 0):
    """Per-node sorted neighbour / weight lists (util_read_data.py:144-187), as numpy int64 arrays."""
    if num_nodes == 0:
        num_nodes = calc_num_nodes_in_mygraph(mygraph)
    n0_to_n1s = [[] for _ in range(num_nodes)]
    n0_to_dts = [[] for _ in range(num_nodes)]
    for n0, n1, dt in mygraph:
        n0_to_n1s[n0].append(n1)
        n0_to_dts[n0].append(dt)
        if if_bidirectional:
            n0_to_n1s[n1].append(n0)
            n0_to_dts[n1].append(dt)
    out_n, out_d = [], []
    for n1s, dts in zip(n0_to_n1s, n0_to_dts):
        a = np.asarray(n1s, dtype=np.int64)
        d = np.asarray(dts, dtype=np.int64)
        order = np.argsort(a, kind="stable")
        out_n.append(a[order])
        out_d.append(d[order])
    return out_n, out_d


def build_adjacency_bool(mygraph: MyGraph, num_nodes: int = 0, if_bidirectional: bool = False) -> np.ndarray:
    """Dense bool adjacency (util.py:343-370).  O(N^2) bytes: only built on request."""
    if num_nodes == 0:
        num_nodes = calc_num_nodes_in_mygraph(mygraph)
    adj = np.zeros((num_nodes, num_nodes), dtype=bool)
    if len(mygraph):
        e = np.asarray([(a, b) for a, b, _ in mygraph], dtype=np.int64)
        adj[e[:, 0], e[:, 1]] = True
    if if_bidirectional:
        adj = np.logical_or(adj, adj.T)
    return adj


# --------------------------------------------------------------------------- #
# device layout
# --------------------------------------------------------------------------- #
@dataclass
class GraphCSR:
    """Host copy of what the kernels read.

    eu/ev/ew : edges exactly as the env stores them (E' = E, or 2E when
               ``if_bidirectional``; ordering = the reference's n0-major,
               n1-sorted order, env_L2A.py:46-48) -- objective kernels.
    rowptr/col/wgt : symmetric CSR over all N nodes, neighbours sorted,
               multi-edges kept, self loops dropped -- delta kernels.
    """
    num_nodes: int
    num_edges: int            # len(mygraph)
    if_bidirectional: bool
    eu: np.ndarray
    ev: np.ndarray
    ew: np.ndarray
    rowptr: np.ndarray
    col: np.ndarray
    wgt: np.ndarray
    max_degree: int

    @property
    def num_stored_edges(self) -> int:
        return int(self.eu.shape[0])

    @property
    def nnz(self) -> int:
        return int(self.col.shape[0])

    @property
    def degree(self) -> np.ndarray:
        return np.diff(self.rowptr)


def build_csr(mygraph_or_arrays, num_nodes: int = 0, if_bidirectional: bool = False) -> GraphCSR:
    """MyGraph (or (eu, ev, w) arrays) -> GraphCSR."""
    if isinstance(mygraph_or_arrays, tuple) and len(mygraph_or_arrays) == 3 and isinstance(
            mygraph_or_arrays[0], np.ndarray):
        u, v, w = (np.asarray(a, dtype=np.int64) for a in mygraph_or_arrays)
    else:
        g = mygraph_or_arrays
        if len(g):
            arr = np.asarray(g, dtype=np.int64).reshape(-1, 3)
            u, v, w = arr[:, 0], arr[:, 1], arr[:, 2]
        else:
            u = v = w = np.zeros(0, np.int64)
    E = int(u.shape[0])
    if num_nodes == 0:
        num_nodes = int(np.unique(np.concatenate([u, v])).shape[0]) if E else 0
    if E and (u.min() < 0 or v.min() < 0 or max(u.max(), v.max()) >= num_nodes):
        raise ValueError("edge endpoint out of range for num_nodes=%d" % num_nodes)
    if E >= 2 ** 31 - 1 or num_nodes >= 2 ** 31 - 1:
        raise ValueError("graph too large for int32 CSR")

    # edges as the env stores them: n0-major, then n1 ascending (stable)
    if if_bidirectional:
        su = np.concatenate([u, v])
        sv = np.concatenate([v, u])
        sw = np.concatenate([w, w])
    else:
        su, sv, sw = u, v, w
    order = np.lexsort((sv, su)) if su.size else np.zeros(0, np.int64)
    eu, ev, ew = su[order], sv[order], sw[order]

    # symmetric CSR (self loops dropped)
    keep = u != v
    cu = np.concatenate([u[keep], v[keep]])
    cv = np.concatenate([v[keep], u[keep]])
    cw = np.concatenate([w[keep], w[keep]])
    o2 = np.lexsort((cv, cu)) if cu.size else np.zeros(0, np.int64)
    cu, cv, cw = cu[o2], cv[o2], cw[o2]
    counts = np.bincount(cu, minlength=num_nodes) if cu.size else np.zeros(num_nodes, np.int64)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return GraphCSR(
        num_nodes=int(num_nodes), num_edges=E, if_bidirectional=bool(if_bidirectional),
        eu=eu.astype(np.int32), ev=ev.astype(np.int32), ew=ew.astype(np.int32),
        rowptr=rowptr.astype(np.int32), col=cv.astype(np.int32), wgt=cw.astype(np.int32),
        max_degree=int(counts.max()) if num_nodes else 0,
    )


# --------------------------------------------------------------------------- #
# deterministic synthetic instances
# --------------------------------------------------------------------------- #
def generate_gnm(num_nodes: int, num_edges: int, seed: int) -> MyGraph:
    """Uniform simple graph G(n, m), weights 1 -- the Gset stand-in (same n, m as G14/G22/G70)."""
    max_e = num_nodes * (num_nodes - 1) // 2
    if num_edges > max_e:
        raise ValueError("too many edges for a simple graph")
    rng = np.random.Generator(np.random.PCG64(seed))
    chosen = np.zeros(0, dtype=np.int64)
    while chosen.shape[0] < num_edges:
        need = num_edges - chosen.shape[0]
        a = rng.integers(0, num_nodes, size=need + need // 4 + 16, dtype=np.int64)
        b = rng.integers(0, num_nodes, size=a.shape[0], dtype=np.int64)
        ok = a != b
        lo, hi = np.minimum(a[ok], b[ok]), np.maximum(a[ok], b[ok])
        chosen = np.concatenate([chosen, lo * num_nodes + hi])
        _, first = np.unique(chosen, return_index=True)
        chosen = chosen[np.sort(first)]
    chosen = np.sort(chosen[:num_edges])
    return [(int(k // num_nodes), int(k % num_nodes), 1) for k in chosen]


def generate_ba(num_nodes: int, m: int, seed: int) -> MyGraph:
    """Barabasi-Albert preferential attachment (the algorithm networkx's
    ``barabasi_albert_graph`` uses, reference util_generate.py:84 with m=4):
    start from a star on m+1 nodes, each new node attaches to m distinct
    targets drawn from the repeated-endpoint list.  E = (n - m - 1) * m + m... exactly
    m*(n-m) for the star start used here."""
    if m < 1 or m >= num_nodes:
        raise ValueError("BA needs 1 <= m < n")
    rng = np.random.Generator(np.random.PCG64(seed))
    edges = [(0, i) for i in range(1, m + 1)]  # star_graph(m): m edges on m+1 nodes
    repeated = [0] * m + list(range(1, m + 1))
    for src in range(m + 1, num_nodes):
        targets = set()
        while len(targets) < m:
            targets.add(repeated[int(rng.integers(0, len(repeated)))])
        for t in sorted(targets):
            edges.append((t, src))
        repeated.extend(targets)
        repeated.extend([src] * m)
    return [(int(min(a, b)), int(max(a, b)), 1) for a, b in edges]


def generate_er(num_nodes: int, p: float, seed: int) -> MyGraph:
    """Erdos-Renyi G(n, p) (reference util_generate.py:80 uses p = 0.15)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    iu, ju = np.triu_indices(num_nodes, k=1)
    keep = rng.random(iu.shape[0]) < p
    return [(int(a), int(b), 1) for a, b in zip(iu[keep], ju[keep])]


def generate_mygraph(graph_type: str, num_nodes: int, seed=None):
    """(mygraph, num_nodes, num_edges), signature of util_generate.py:77-93."""
    graph_type = getattr(graph_type, "value", graph_type)
    seed = 0 if seed is None else seed
    if graph_type == "ER":
        g = generate_er(num_nodes, 0.15, seed)
    elif graph_type == "BA":
        g = generate_ba(num_nodes, 4, seed)
    elif graph_type == "PL":
        # powerlaw_cluster(m=4, p=0.05) ~ BA with occasional triad closure; the
        # env kernels only need a graph of that density, so BA(m=4) stands in.
        g = generate_ba(num_nodes, 4, seed + 7919)
    else:
        raise ValueError(f"g_type {graph_type} should in ['ER', 'PL', 'BA']")
    return g, num_nodes, len(g)


GSET_SIZES = {14: (800, 4694), 15: (800, 4661), 22: (2000, 19990), 49: (3000, 6000),
              50: (3000, 6000), 55: (5000, 12498), 70: (10000, 9999)}


def load_gset(k: int, data_dir: str = "data/gset"):
    """Real ``gset_<k>.txt`` if the user supplied it, else the same-(n, m) G(n, m)
    stand-in with seed k.  Returns (mygraph, num_nodes, is_real)."""
    p = os.path.join(data_dir, f"gset_{k}.txt")
    if os.path.exists(p):
        n, m = read_graph_header(p)
        if (n, m) == GSET_SIZES.get(k, (n, m)):
            return read_mygraph(p), n, True
    n, m = GSET_SIZES[k]
    return generate_gnm(n, m, seed=k), n, False


# --------------------------------------------------------------------------- #
# TSP instances
# --------------------------------------------------------------------------- #
def read_tsp_file(file_path: str) -> List[List[float]]:
    """TSPLIB NODE_COORD_SECTION reader (ISCO/util_TSP.py:27-45)."""
    cities = []
    with open(file_path, "r") as f:
        start = False
        for line in f:
            s = line.strip()
            if s == "NODE_COORD_SECTION":
                start = True
                continue
            if start:
                if s == "EOF" or not s:
                    break
                p = s.split()
                cities.append([float(p[1]), float(p[2])])
    return cities


def tsp_tables(coords: np.ndarray, K: int = 20):
    """distance f32 [N,N], nearest_indices int64 [N,K], random_indices int64 [N,N-1]
    with the semantics of ISCO/util_TSP.py:5-23 (cdist p=2 in f32; K nearest
    excluding self, ascending; random_indices[i] = all cities but i, ascending)."""
    c = np.asarray(coords, dtype=np.float32)
    n = c.shape[0]
    diff = c[:, None, :].astype(np.float32) - c[None, :, :].astype(np.float32)
    dist = np.sqrt((diff * diff).sum(-1, dtype=np.float32)).astype(np.float32)
    k = min(K, n - 1)
    order = np.argsort(dist, axis=1, kind="stable")
    nearest = np.empty((n, k), dtype=np.int64)
    for i in range(n):
        row = order[i]
        row = row[row != i][:k] if row[0] != i else row[1:k + 1]
        nearest[i] = row
    rnd = np.empty((n, n - 1), dtype=np.int64)
    ar = np.arange(n)
    for i in range(n):
        rnd[i] = np.concatenate([ar[:i], ar[i + 1:]])
    return dist, nearest, rnd


def load_data(tsp_file_path: str, K: int = 20, device=None) -> dict:
    """ISCO/util_TSP.py:5-23: a TSPLIB file -> the ``params_dict`` ISCO_TSP is built from (``distance`` f32 [N, N],
    ``nearest_indices`` int64 [N, K], ``random_indices`` int64 [N, N-1], ``num_nodes``) as torch tensors.  K and the
    device are star-imported config constants there, arguments here."""
    import torch
    dist, nearest, rnd = tsp_tables(np.asarray(read_tsp_file(tsp_file_path), dtype=np.float32), K)
    put = (lambda a: torch.from_numpy(a).to(device)) if device is not None else torch.from_numpy
    return {"distance": put(dist), "nearest_indices": put(nearest), "random_indices": put(rnd), "num_nodes": int(dist.shape[0])}


load_tsp = load_data


def generate_tsp_coords(num_nodes: int, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random((num_nodes, 2), dtype=np.float64).astype(np.float32)
