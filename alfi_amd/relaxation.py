"""Patch construction for the level smoother -- same protocol and class names as alfi/relaxation.py.

A patch constructor is a zero-argument-constructible class whose ``__call__(pc)`` returns ``(patches, iterset)``:
``patches`` is a list of arrays of mesh points, ``iterset`` the order in which patches are visited
(relaxation.py:110-150).  The reference talks to a PETSc DMPlex; here the same queries (``getTransitiveClosure``,
``getDepthStratum``, ``getLabelValue`` ...) are answered by ``PlexLike``, a thin view of ``alfi_amd.mesh.SimplexMesh``
with DMPlex's point numbering convention (cells, then vertices, then edges, then faces), and options come from an
``Options`` object with PETSc.Options' ``getInt`` / ``getString`` (relaxation.py:78-108, 112-118).

The built-in ``patch_pc_patch_construct_type: star`` (solver.py:337-338) does not go through this module: it uses the
vectorised ``VectorFunctionSpace.star_patches``; ``tests/test_frontend.py`` checks that both agree.
"""
import itertools

import numpy as np


class Options(object):
    """Stand-in for PETSc.Options(prefix) over a flat dict (keys without the leading dash)."""

    def __init__(self, prefix="", entries=None):
        self.prefix = prefix or ""
        self.entries = entries if entries is not None else {}

    def _get(self, name, default):
        return self.entries.get(self.prefix + name, default)

    def getInt(self, name, default=None):
        v = self._get(name, default)
        return v if v is default else int(v)

    def getString(self, name, default=None):
        v = self._get(name, default)
        return v if v is default else str(v)

    def getBool(self, name, default=None):
        v = self._get(name, default)
        if v is default:
            return v
        return v if isinstance(v, bool) else str(v).lower() in ("1", "true", "yes")


class PlexLike(object):
    """DMPlex-style topology queries on a SimplexMesh.  Points: cells [0, nc), vertices, edges, (faces)."""

    def __init__(self, mesh, labels=None):
        self.mesh = mesh
        nc, nv, ne, nf = mesh.num_cells, mesh.num_vertices, mesh.num_edges, mesh.num_faces
        self.cStart, self.vStart, self.eStart = 0, nc, nc + nv
        self.fStart = nc + nv + ne
        self.pEnd = nc + nv + ne + nf
        self.labels = labels or {}
        self._support = None

    def getDimension(self):
        return self.mesh.dim

    def getDepthStratum(self, depth):
        m = self.mesh
        bounds = {0: (self.vStart, self.eStart), 1: (self.eStart, self.fStart),
                  m.dim: (self.cStart, self.vStart)}
        if m.dim == 3:
            bounds[2] = (self.fStart, self.pEnd)
        return bounds[depth]

    def getHeightStratum(self, height):
        return self.getDepthStratum(self.mesh.dim - height)

    def getCone(self, p):
        m = self.mesh
        if p < self.vStart:                                   # cell -> facets
            if m.dim == 3:
                return self.fStart + m.cell_faces[p]
            return self.eStart + m.cell_edges[p]
        if p < self.eStart:
            return np.array([], dtype=np.int64)
        if p < self.fStart:                                   # edge -> vertices
            return self.vStart + m.edges[p - self.eStart]
        f = m.faces[p - self.fStart]                          # face -> edges
        es = []
        for a, b in ((0, 1), (0, 2), (1, 2)):
            key = (min(f[a], f[b]), max(f[a], f[b]))
            es.append(self._edge_lookup()[key])
        return self.eStart + np.array(es)

    def _edge_lookup(self):
        if not hasattr(self, "_elook"):
            self._elook = {(int(a), int(b)): i for i, (a, b) in enumerate(self.mesh.edges)}
        return self._elook

    def _build_support(self):
        sup = [[] for _ in range(self.pEnd)]
        for p in range(self.pEnd):
            for q in self.getCone(p):
                sup[int(q)].append(p)
        self._support = sup

    def getSupport(self, p):
        if self._support is None:
            self._build_support()
            self._support_arr = {}
        a = self._support_arr.get(p)
        if a is None:
            a = self._support_arr[p] = np.array(self._support[p], dtype=np.int64)
        return a

    def getTransitiveClosure(self, p, useCone=True):
        # memoised: a macro-star constructor asks for the star of the same point from every patch that contains it
        # (2 M star requests for 0.3 M distinct points on the 441 k-dof channel); the order of the points is DMPlex's
        # breadth-first one either way
        cache = self.__dict__.setdefault("_closure_cache", ({}, {}))[0 if useCone else 1]
        p = int(p)
        hit = cache.get(p)
        if hit is not None:
            return hit, None
        seen, mark, frontier = [p], {p}, [p]
        nxt = self.getCone if useCone else self.getSupport
        while frontier:
            new = []
            for q in frontier:
                for r in nxt(q):
                    r = int(r)
                    if r not in mark:
                        mark.add(r)
                        seen.append(r)
                        new.append(r)
            frontier = new
        out = cache[p] = np.array(seen, dtype=np.int64)
        return out, None

    def getLabelValue(self, name, p):
        lab = self.labels.get(name)
        if lab is None:
            return -1
        return int(lab.get(int(p), -1)) if isinstance(lab, dict) else int(lab[int(p)])

    def point_coords(self, p):
        """Mean of the vertex coordinates in the closure of p (relaxation.py:61-67)."""
        m = self.mesh
        if p < self.vStart:
            return m.coords[m.cells[p]].mean(axis=0)
        if p < self.eStart:
            return m.coords[p - self.vStart]
        if p < self.fStart:
            return m.coords[m.edges[p - self.eStart]].mean(axis=0)
        return m.coords[m.faces[p - self.fStart]].mean(axis=0)


def select_entity(p, dm=None, exclude=None):
    """relaxation.py:8-19."""
    if exclude is None:
        return True
    return dm.getLabelValue(exclude, p) == -1


class OrderedRelaxation(object):
    """relaxation.py:22-150."""

    def __init__(self):
        self.name = None

    def callback(self, dm, entity):
        raise NotImplementedError

    def set_options(self, dm, opts, name):
        pass

    @staticmethod
    def star(dm, p):
        return dm.getTransitiveClosure(p, useCone=False)[0]

    @staticmethod
    def closure(dm, p):
        return dm.getTransitiveClosure(p, useCone=True)[0]

    @staticmethod
    def cone(dm, p):
        return dm.getCone(p)

    @staticmethod
    def support(dm, p):
        return dm.getSupport(p)

    @staticmethod
    def coords(dm, p):
        return dm.point_coords(p)

    @staticmethod
    def get_entities(opts, name, dm):
        sentinel = object()
        codim = opts.getInt("pc_patch_construction_%s_codim" % name, default=sentinel)
        if codim is sentinel:
            dim = opts.getInt("pc_patch_construction_%s_dim" % name, default=0)
            return range(*dm.getDepthStratum(dim))
        return range(*dm.getHeightStratum(codim))

    @staticmethod
    def parse_sort_order(text):
        """``sort_order`` grammar of relaxation.py:88-108: sweeps separated by '|', each sweep a ':'-separated list of
        ``axis[+|-]`` giving a lexicographic key.  Returns a list of sweeps, each a list of (axis, sign)."""
        if text is None or text in ("None", ""):
            return None
        sweeps = []
        for sweep in text.split("|"):
            keys = []
            for tok in sweep.split(":"):
                sign = -1 if tok[1:2] == "-" else 1
                if len(tok) > 1 and tok[1] not in "+-":
                    raise ValueError("bad sort_order token %r" % tok)
                keys.append((int(tok[0]), sign))
            sweeps.append(keys)
        return sweeps

    def iteration_order(self, centroids):
        """Concatenation of one coordinate-sorted permutation per sweep (relaxation.py:139-150), or the identity.

        With several '|'-separated sweeps the reference's key functions are closures over the loop variable ``sortdata``
        (relaxation.py:98-108), so at sort time EVERY sweep uses the key of the LAST one: 'a|b' yields the b-ordering twice.
        That literal behaviour is the default here (a drop-in must visit the patches in the same order);
        ``pc_patch_construction_<Name>_sort_order_per_sweep: True`` gives every sweep its own key instead.  The shipped
        problems use a single sweep ('0+:1-', examples/ldc2d/ldc2d.py:39), where both agree."""
        sweeps = self.parse_sort_order(self.opts.getString("pc_patch_construction_%s_sort_order" % self.name,
                                                           default=None))
        n = len(centroids)
        if sweeps is None:
            return np.arange(n, dtype=np.int64)
        if not self.opts.getBool("pc_patch_construction_%s_sort_order_per_sweep" % self.name, default=False):
            sweeps = [sweeps[-1]] * len(sweeps)
        X = np.asarray(centroids, dtype=np.float64).reshape(n, -1)
        order = []
        for keys in sweeps:
            # np.lexsort sorts by the LAST key first; ties keep the original order like Python's stable sorted()
            cols = [sign * X[:, ax] for (ax, sign) in reversed(keys)]
            order.append(np.lexsort(cols))
        return np.concatenate(order).astype(np.int64)

    def __call__(self, pc):
        assert self.name is not None
        dm = pc.getDM()
        self.opts = Options(pc.getOptionsPrefix(), getattr(pc, "options", None))
        self.set_options(dm, self.opts, self.name)
        patches, seeds = [], []
        for entity in self.get_entities(self.opts, self.name, dm):
            if not select_entity(entity, dm=dm, exclude="pyop2_ghost"):      # owned entities only (:120-121)
                continue
            points = self.callback(dm, entity)
            if points is None:
                continue
            patches.append(np.asarray(points, dtype=np.int64))
            seeds.append(entity)
        centroids = [self.coords(dm, p) for p in seeds]
        self.seeds = seeds                      # seed entity of every patch (not in the reference; the partitioner wants it)
        return patches, self.iteration_order(centroids)


class Star(OrderedRelaxation):
    """relaxation.py:153-160."""

    def __init__(self):
        super().__init__()
        self.name = "Star"

    def callback(self, dm, vertex):
        return list(self.star(dm, vertex))


class MacroStar(OrderedRelaxation):
    """relaxation.py:163-177 (needs the ``MacroVertices`` label of a barycentrically refined mesh, bary.py:18-19)."""

    def __init__(self):
        super().__init__()
        self.name = "MacroStar"

    def callback(self, dm, vertex):
        if dm.getLabelValue("MacroVertices", vertex) != 1:
            return None
        s = list(self.star(dm, vertex))
        # (concatenations written with chain: the same lists as sum(..., []) without its quadratic copying)
        closures = list(itertools.chain.from_iterable(self.closure(dm, e) for e in s))
        # literal: every closure point whose label is not 1 (relaxation.py:174), not only vertices
        the_vertices_we_care_about = [v for v in closures if dm.getLabelValue("MacroVertices", v) != 1]
        their_star = list(itertools.chain.from_iterable(self.star(dm, v) for v in the_vertices_we_care_about))
        return s + their_star


def patch_points_to_dofs(V, dm, patches):
    """What PCPATCH does with a constructor's point lists [3P]: collect the dofs living on the listed points, drop
    Dirichlet dofs and duplicates.  Returns (patch_ptr, patch_dofs) with ascending dofs per patch; empty patches are
    dropped (the returned ``kept`` lists the surviving patch indices)."""
    d = V.dim
    ptr, dofs, kept = [0], [], []
    for i, pts in enumerate(patches):
        nodes = []
        for p in np.unique(pts):
            p = int(p)
            if dm.vStart <= p < dm.eStart:
                nodes.append(int(V.vertex_nodes[p - dm.vStart]))
            elif dm.eStart <= p < dm.fStart and V.element.has_edge_nodes:
                nodes.extend(int(q) for q in np.atleast_1d(V.edge_nodes[p - dm.eStart]))
            elif p >= dm.fStart and V.element.has_face_nodes:
                nodes.append(int(V.face_nodes[p - dm.fStart]))
        nodes = sorted(n for n in set(nodes) if not V.bc_node_mask[n])
        if not nodes:
            continue
        kept.append(i)
        dofs.extend(n * d + c for n in nodes for c in range(d))
        ptr.append(len(dofs))
    return np.array(ptr, dtype=np.int64), np.array(dofs, dtype=np.int32), kept
