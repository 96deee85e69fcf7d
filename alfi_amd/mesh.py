"""Simplicial meshes and uniformly refined hierarchies for the synthetic ldc2d / ldc3d workloads.

The reference gets its meshes from Firedrake/DMPlex (``RectangleMesh(N, N, 2, 2, diagonal="left")``,
examples/ldc2d/ldc2d.py:16-20; ``BoxMesh(N, N, N, 2, 2, 2)``, examples/ldc3d/ldc3d.py:12-15) and its hierarchy from
``MeshHierarchy`` with the ``prolongation`` facet label (alfi/solver.py:101-110).  None of that is available
here, so this module provides the same geometric objects from scratch:

* ``rectangle_mesh`` / ``box_mesh``: structured triangulations ("left" diagonals in 2-D, 6 Kuhn tets per cube in 3-D).
* ``refine``: regular (red / Bey) refinement with explicit parent->child maps.  Every fine vertex records the (one or
  two) coarse vertices it was created from; from that the "coarse skeleton" membership of every fine entity -- the
  information the reference stores in the ``prolongation`` label (alfi/transfer.py:36-38, 131-133) -- follows by counting
  distinct coarse support vertices (an entity lies inside a coarse facet iff its support has <= dim vertices).

All index arrays are int32 (the reference runs PETSc with 32-bit indices, examples/submission_template.pbs:28).
"""
import numpy as np

# local edges of a simplex, as pairs of local vertices
TRI_EDGES = np.array([(1, 2), (0, 2), (0, 1)], dtype=np.int32)            # edge i is opposite vertex i
TET_EDGES = np.array([(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)], dtype=np.int32)
TET_FACES = np.array([(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)], dtype=np.int32)  # face i is opposite vertex i


def _unique_rows(rows, nv):
    """Unique sorted vertex tuples -> (unique rows, inverse).  rows: (m, k) with sorted entries along axis 1."""
    key = rows[:, 0].astype(np.int64)
    for j in range(1, rows.shape[1]):
        key = key * nv + rows[:, j]
    # (no return_index: np.unique then sorts with the vectorised quicksort instead of a stable merge sort; the rows are read
    # back out of the unique keys)
    ukey, inv = np.unique(key, return_inverse=True)
    k = rows.shape[1]
    out = np.empty((ukey.shape[0], k), dtype=np.int32)
    for j in range(k - 1, 0, -1):
        out[:, j] = ukey % nv
        ukey = ukey // nv
    out[:, 0] = ukey
    return out, inv.astype(np.int32)


class SimplexMesh(object):
    """coords: (nv, dim) float64; cells: (nc, dim+1) int32 (local vertex order is significant for refinement)."""

    def __init__(self, coords, cells, vertex_parents=None, parent_cell=None, child_index=None):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.dim = self.coords.shape[1]
        assert self.cells.shape[1] == self.dim + 1
        # refinement provenance (None on the base mesh)
        self.vertex_parents = vertex_parents   # (nv, 2) coarse vertices each fine vertex was created from
        self.parent_cell = parent_cell         # (nc,) coarse cell of each cell
        self.child_index = child_index         # (nc,) which child (0..2^dim-1) of its parent
        self._build_entities()

    # -- topology ---------------------------------------------------------------------------------------------
    def _build_entities(self):
        nv = self.coords.shape[0]
        c = self.cells
        le = TRI_EDGES if self.dim == 2 else TET_EDGES
        e = np.sort(c[:, le].reshape(-1, 2), axis=1)
        self.edges, inv = _unique_rows(e, nv)
        self.cell_edges = inv.reshape(c.shape[0], le.shape[0])
        if self.dim == 3:
            f = np.sort(c[:, TET_FACES].reshape(-1, 3), axis=1)
            self.faces, inv = _unique_rows(f, nv)
            self.cell_faces = inv.reshape(c.shape[0], 4)
            self.facets, self.cell_facets = self.faces, self.cell_faces
        else:
            self.faces, self.cell_faces = None, None
            self.facets, self.cell_facets = self.edges, self.cell_edges
        cnt = np.bincount(self.cell_facets.ravel(), minlength=self.facets.shape[0])
        self.boundary_facets = np.flatnonzero(cnt == 1).astype(np.int32)

    @property
    def num_vertices(self):
        return self.coords.shape[0]

    @property
    def num_cells(self):
        return self.cells.shape[0]

    @property
    def num_edges(self):
        return self.edges.shape[0]

    @property
    def num_faces(self):
        return 0 if self.faces is None else self.faces.shape[0]

    def boundary_entities(self, select=None):
        """(vertex mask, edge mask, face mask) of entities in the closure of exterior facets; ``select``: callable on the
        facet centroids (nbf, dim) -> bool mask, restricts to a part of the boundary (DirichletBC on some labels only)."""
        vm = np.zeros(self.num_vertices, dtype=bool)
        em = np.zeros(self.num_edges, dtype=bool)
        fm = np.zeros(self.num_faces, dtype=bool)
        bf = self.boundary_facets
        if select is not None:
            bf = bf[np.asarray(select(self.coords[self.facets[bf]].mean(axis=1)), dtype=bool)]
        vm[self.facets[bf].ravel()] = True
        if self.dim == 2:
            em[bf] = True
        else:
            fm[bf] = True
            # edges of boundary faces: find cells owning each boundary face and take the 3 edges of that face
            cf = self.cell_faces
            isb = np.zeros(self.num_faces, dtype=bool)
            isb[bf] = True
            ci, li = np.nonzero(isb[cf])
            # edges of local face li: the local edges not touching vertex li
            face_edges = np.array([[j for j, (a, b) in enumerate(TET_EDGES) if a != i and b != i] for i in range(4)])
            em[self.cell_edges[ci[:, None], face_edges[li]].ravel()] = True
        return vm, em, fm

    def cell_geometry(self):
        """Barycentric-coordinate gradients g (nc, dim+1, dim) and cell volumes (nc,); computed once per mesh."""
        if getattr(self, "_geometry", None) is None:
            self._geometry = self._cell_geometry()
        return self._geometry

    def _cell_geometry(self):
        x = self.coords[self.cells]                       # (nc, d+1, d)
        J = (x[:, 1:, :] - x[:, :1, :]).transpose(0, 2, 1)  # columns = edge vectors
        det = np.linalg.det(J)
        Jinv = np.linalg.inv(J)                           # rows of Jinv = grad lambda_1..d
        g = np.empty((self.num_cells, self.dim + 1, self.dim))
        g[:, 1:, :] = Jinv
        g[:, 0, :] = -Jinv.sum(axis=1)
        fact = 2.0 if self.dim == 2 else 6.0
        return g, np.abs(det) / fact


def rectangle_mesh(nx, ny, Lx, Ly, diagonal="left"):
    """Structured triangulation of [0,Lx]x[0,Ly] (ldc2d.py:16-20; reference default diagonal is "left")."""
    xs, ys = np.linspace(0, Lx, nx + 1), np.linspace(0, Ly, ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    v00 = (j * (nx + 1) + i).ravel()
    v10, v01, v11 = v00 + 1, v00 + nx + 1, v00 + nx + 2
    if diagonal == "left":      # diagonal from (i, j+1) to (i+1, j)
        t0 = np.stack([v00, v10, v01], axis=1)
        t1 = np.stack([v10, v11, v01], axis=1)
    elif diagonal == "right":   # diagonal from (i, j) to (i+1, j+1)
        t0 = np.stack([v00, v10, v11], axis=1)
        t1 = np.stack([v00, v11, v01], axis=1)
    else:
        raise NotImplementedError("diagonal %r" % diagonal)
    cells = np.stack([t0, t1], axis=1).reshape(-1, 3)
    return SimplexMesh(coords, cells)


def box_mesh(nx, ny, nz, Lx, Ly, Lz):
    """Structured tetrahedralisation of a box, 6 Kuhn tets per cube sharing the main diagonal (ldc3d.py:12-15)."""
    from itertools import permutations
    xs, ys, zs = np.linspace(0, Lx, nx + 1), np.linspace(0, Ly, ny + 1), np.linspace(0, Lz, nz + 1)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    stride = np.array([1, nx + 1, (nx + 1) * (ny + 1)], dtype=np.int64)
    base = (k * stride[2] + j * stride[1] + i).ravel()
    tets = []
    for perm in permutations(range(3)):
        p0 = base
        p1 = p0 + stride[perm[0]]
        p2 = p1 + stride[perm[1]]
        p3 = p2 + stride[perm[2]]
        tets.append(np.stack([p0, p1, p2, p3], axis=1))
    cells = np.stack(tets, axis=1).reshape(-1, 4)
    return SimplexMesh(coords, cells)


# children of a simplex in terms of (parent local vertex a, parent local vertex b): the child's vertex is the midpoint
# of a and b (a == b: the parent vertex itself).  3-D: Bey's rule, which maps Kuhn tets to Kuhn tets.
_TRI_CHILDREN = [
    [(0, 0), (0, 1), (0, 2)],
    [(0, 1), (1, 1), (1, 2)],
    [(0, 2), (1, 2), (2, 2)],
    [(1, 2), (0, 2), (0, 1)],
]
_TET_CHILDREN = [
    [(0, 0), (0, 1), (0, 2), (0, 3)],
    [(0, 1), (1, 1), (1, 2), (1, 3)],
    [(0, 2), (1, 2), (2, 2), (2, 3)],
    [(0, 3), (1, 3), (2, 3), (3, 3)],
    [(0, 1), (0, 2), (0, 3), (1, 3)],
    [(0, 1), (0, 2), (1, 2), (1, 3)],
    [(0, 2), (0, 3), (1, 3), (2, 3)],
    [(0, 2), (1, 2), (1, 3), (2, 3)],
]


def children_table(dim):
    return _TRI_CHILDREN if dim == 2 else _TET_CHILDREN


def child_barycentric(dim):
    """B[k]: (dim+1, dim+1) parent-barycentric coordinates of the vertices of child k."""
    tab = children_table(dim)
    B = np.zeros((len(tab), dim + 1, dim + 1))
    for k, child in enumerate(tab):
        for r, (a, b) in enumerate(child):
            B[k, r, a] += 0.5
            B[k, r, b] += 0.5
    return B


def refine(mesh):
    """Regular refinement: every cell -> 2^dim children; child k of coarse cell c is fine cell c*2^dim + k."""
    dim = mesh.dim
    nv = mesh.num_vertices
    le = TRI_EDGES if dim == 2 else TET_EDGES
    # local (a,b) -> local edge index
    emap = -np.ones((dim + 1, dim + 1), dtype=np.int64)
    for j, (a, b) in enumerate(le):
        emap[a, b] = emap[b, a] = j
    mid = nv + mesh.cell_edges.astype(np.int64)          # (nc, nedges) vertex id of each edge midpoint
    coords = np.concatenate([mesh.coords, 0.5 * (mesh.coords[mesh.edges[:, 0]] + mesh.coords[mesh.edges[:, 1]])])
    vparents = np.concatenate([np.repeat(np.arange(nv, dtype=np.int32)[:, None], 2, axis=1), mesh.edges])
    tab = children_table(dim)
    nch = len(tab)
    cells = np.empty((mesh.num_cells, nch, dim + 1), dtype=np.int64)
    for k, child in enumerate(tab):
        for r, (a, b) in enumerate(child):
            cells[:, k, r] = mesh.cells[:, a] if a == b else mid[:, emap[a, b]]
    cells = cells.reshape(-1, dim + 1)
    parent = np.repeat(np.arange(mesh.num_cells, dtype=np.int32), nch)
    cidx = np.tile(np.arange(nch, dtype=np.int32), mesh.num_cells)
    return SimplexMesh(coords, cells, vertex_parents=vparents.astype(np.int32), parent_cell=parent, child_index=cidx)


def mesh_hierarchy(base, nref):
    """[base, refine(base), ...] -- the analogue of firedrake.MeshHierarchy(base, nref) (alfi/problem.py:10-24)."""
    mh = [base]
    for _ in range(nref):
        mh.append(refine(mh[-1]))
    return mh


def coarse_support_size(mesh, entity_vertices):
    """Number of distinct coarse vertices in the union of the parents of the given fine vertices.

    entity_vertices: (m, k) fine vertex ids.  An entity lies on the closure of a coarse facet (the coarse skeleton,
    alfi/transfer.py:121-158) iff this number is <= dim; it is interior to a coarse cell iff it equals dim+1."""
    par = mesh.vertex_parents[entity_vertices].reshape(entity_vertices.shape[0], -1)   # (m, 2k)
    par = np.sort(par, axis=1)
    return 1 + (np.diff(par, axis=1) != 0).sum(axis=1)


# -- barycentric (Alfeld) refinement: the meshes of the reference's Scott-Vogelius discretisation ------------------------
def bary_refine(mesh):
    """Alfeld split (PETSc DMPlexTransform REFINEALFELD, alfi/bary.py:16-27): every cell gets its barycentre as a new
    vertex and is replaced by dim+1 cells, child i = the parent with vertex i moved to the barycentre; fine cell
    c*(dim+1)+i is child i of cell c (the numbering bary.py:148-151 relies on).  The vertices of the mesh that is split
    are the macro vertices (``MacroVertices`` label = 1, bary.py:18-19): ``macro_vertex_mask`` on the result."""
    dim, nv, nc = mesh.dim, mesh.num_vertices, mesh.num_cells
    centres = mesh.coords[mesh.cells].mean(axis=1)
    coords = np.concatenate([mesh.coords, centres])
    cells = np.repeat(mesh.cells.astype(np.int64)[:, None, :], dim + 1, axis=1)          # (nc, dim+1, dim+1)
    for i in range(dim + 1):
        cells[:, i, i] = nv + np.arange(nc)
    out = SimplexMesh(coords, cells.reshape(-1, dim + 1),
                      parent_cell=np.repeat(np.arange(nc, dtype=np.int32), dim + 1),
                      child_index=np.tile(np.arange(dim + 1, dtype=np.int32), nc))
    out.macro_vertex_mask = np.concatenate([np.ones(nv, dtype=bool), np.zeros(nc, dtype=bool)])
    out.macro_mesh = mesh
    return out


def bary_mesh_hierarchy(base, nref):
    """alfi.bary.BaryMeshHierarchy (bary.py:29-194): level l = Alfeld split of the l-times uniformly refined base mesh.
    The levels are NOT nested in one another (only their macro meshes are); ``macro_mesh`` on every level gives the
    nested skeleton the reference's coarse_to_fine maps are built from (bary.py:137-160)."""
    return [bary_refine(m) for m in mesh_hierarchy(base, nref)]


def bfs3d_mesh(n):
    """Structured stand-in for the reference's gmsh backward-facing-step channel (examples/bfs3d/
    backwards-facing-step-3d.geo:1-6: a 10 x 2 x 1 channel with a 1 x 1 step under the inlet): Kuhn tets on n cubes per
    unit length, the cubes of the step [0, 1] x [0, 1] x [0, 1] removed."""
    full = box_mesh(10 * n, 2 * n, n, 10.0, 2.0, 1.0)
    centre = full.coords[full.cells].mean(axis=1)
    keep = ~((centre[:, 0] < 1.0) & (centre[:, 1] < 1.0))
    cells = full.cells[keep]
    used = np.unique(cells)
    renum = np.full(full.num_vertices, -1, dtype=np.int64)
    renum[used] = np.arange(used.shape[0])
    return SimplexMesh(full.coords[used], renum[cells])


# -- gmsh 2.2 ASCII meshes: the format of the reference's channel meshes (examples/bfs3d/coarse*.msh, written by gmsh from
#    backwards-facing-step-3d.geo; firedrake.Mesh reads them, bfs3d.py:13-16) -----------------------------------------------
def read_gmsh(path):
    """Simplicial mesh from a gmsh ``$MeshFormat 2.2 0 8`` ASCII file: ``$Nodes`` (id x y z) and ``$Elements`` (id type
    ntags tags... nodes); tetrahedra (type 4) / triangles (type 2) are the cells in 3-D / 2-D, the elements one dimension
    lower (triangles / lines, type 1) carry the physical tag of the boundary they lie on -- kept in
    ``mesh.boundary_tags``: {sorted vertex tuple of the facet: physical tag} (bfs3d.py:23-26 uses 1 = inflow, 3 = walls;
    2 = outflow stays natural)."""
    with open(path) as fh:
        lines = fh.read().split("\n")
    pos = {l.strip(): i for i, l in enumerate(lines) if l.startswith("$")}
    if "$MeshFormat" not in pos or not lines[pos["$MeshFormat"] + 1].startswith("2."):
        raise ValueError("%s: not a gmsh 2.x ASCII mesh" % path)
    if lines[pos["$MeshFormat"] + 1].split()[1] != "0":
        raise ValueError("%s: binary gmsh files are not supported" % path)
    n0 = pos["$Nodes"]
    nn = int(lines[n0 + 1])
    tab = np.array([l.split() for l in lines[n0 + 2:n0 + 2 + nn]], dtype=np.float64)
    ids = tab[:, 0].astype(np.int64)
    renum = np.full(ids.max() + 1, -1, dtype=np.int64)
    renum[ids] = np.arange(nn)
    xyz = tab[:, 1:4]
    e0 = pos["$Elements"]
    ne = int(lines[e0 + 1])
    by_type = {}
    for l in lines[e0 + 2:e0 + 2 + ne]:
        t = l.split()
        et, ntags = int(t[1]), int(t[2])
        by_type.setdefault(et, []).append((int(t[3]) if ntags else 0, [int(v) for v in t[3 + ntags:]]))
    nv_of = {1: 2, 2: 3, 4: 4, 15: 1}
    dim = 3 if 4 in by_type else 2
    cell_t, facet_t = (4, 2) if dim == 3 else (2, 1)
    if cell_t not in by_type:
        raise ValueError("%s holds no simplicial cells" % path)
    cells = renum[np.array([v[:nv_of[cell_t]] for _, v in by_type[cell_t]], dtype=np.int64)]
    used = np.unique(cells)
    compact = np.full(nn, -1, dtype=np.int64)
    compact[used] = np.arange(used.shape[0])
    coords = xyz[used][:, :dim] if dim == 2 and np.abs(xyz[used][:, 2]).max() == 0.0 else xyz[used]
    mesh = SimplexMesh(coords[:, :dim] if dim == 2 else coords, compact[cells])
    mesh.boundary_tags = {}
    for tag, v in by_type.get(facet_t, []):
        key = tuple(sorted(int(compact[renum[q]]) for q in v[:nv_of[facet_t]]))
        if min(key) >= 0:
            mesh.boundary_tags[key] = tag
    return mesh


def write_gmsh(mesh, path, tag_of_facet=None):
    """The inverse of ``read_gmsh`` for this package's own meshes (tests, exchanging meshes with gmsh-based tools):
    ``tag_of_facet``: callable on the boundary facet centroids (nbf, dim) -> integer physical tags (default 1)."""
    dim = mesh.dim
    bf = mesh.boundary_facets
    cent = mesh.coords[mesh.facets[bf]].mean(axis=1)
    tags = np.ones(len(bf), dtype=np.int64) if tag_of_facet is None else np.asarray(tag_of_facet(cent), dtype=np.int64)
    with open(path, "w") as fh:
        fh.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % mesh.num_vertices)
        for i, x in enumerate(mesh.coords):
            xyz = list(x) + [0.0] * (3 - dim)
            fh.write("%d %.16g %.16g %.16g\n" % (i + 1, xyz[0], xyz[1], xyz[2]))
        fh.write("$EndNodes\n$Elements\n%d\n" % (len(bf) + mesh.num_cells))
        k = 1
        ft, ct = (2, 4) if dim == 3 else (1, 2)
        for f, t in zip(bf, tags):
            fh.write("%d %d 2 %d %d %s\n" % (k, ft, t, t, " ".join(str(int(v) + 1) for v in mesh.facets[f])))
            k += 1
        for c in mesh.cells:
            fh.write("%d %d 2 1 1 %s\n" % (k, ct, " ".join(str(int(v) + 1) for v in c)))
            k += 1
        fh.write("$EndElements\n")
