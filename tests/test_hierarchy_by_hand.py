"""The generator's refinement, node sets, vertex stars and coarse-cell interior blocks on the smallest mesh against a table
enumerated BY HAND from the definitions (no alfi_amd routine on the expected side).  Oracle and product share
alfi_amd.problem.build_hierarchy, so a numbering or refinement bug there would be invisible to the parity tests: this pins it.
CPU only."""
import numpy as np


def test_hierarchy_tables_on_the_smallest_meshes():
    """build_hierarchy's refinement, node sets, vertex stars and coarse-cell interior blocks on the N = 1 unit-square-type mesh
    (2 triangles -> 8 triangles) against a table enumerated BY HAND below from the definitions -- no alfi_amd routine on the
    expected side.  Reference: alfi/relaxation.py:110-150 (star of a vertex = the dofs in the interior of the cells around it),
    alfi/transfer.py:13-46 (interior dofs of a coarse cell's children), :121-158 (dofs on the coarse skeleton are excluded)."""
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, build_hierarchy
    lv, tr = build_hierarchy(TwoDimLidDrivenCavityProblem(1), 1, 2, Re=0.0, gamma=1.0)
    Vc, Vf = lv[0].V, lv[1].V
    # the domain is [0, 2]^2 split into 2 triangles by ONE diagonal; red refinement gives 8 triangles on a 3 x 3 vertex lattice
    # (spacing 1) and, with P2, nodes on the 5 x 5 lattice of spacing 1/2 except the lattice points that are neither a vertex
    # nor an edge midpoint of the refined triangulation
    assert Vc.mesh.num_cells == 2 and Vf.mesh.num_cells == 8
    assert Vc.mesh.num_vertices == 4 and Vf.mesh.num_vertices == 9
    # P2 nodes = vertices + edges: coarse 4 + 5 = 9, fine 9 + 16 = 25 (Euler: V - E + F = 1 -> E = 9 + 8 - 1 = 16)
    assert Vc.num_nodes == 9 and Vf.num_nodes == 25
    pts = {tuple(np.round(p * 2).astype(int)) for p in Vf.node_coords}
    assert pts == {(i, j) for i in range(5) for j in range(5)}              # every point of the half-spacing lattice is a node
    # which coarse diagonal? the one whose midpoint (1, 1) is a coarse node shared by both cells: either, (1,1) is on both
    diag = [c for c in Vc.mesh.cells if True]
    shared = set(diag[0]) & set(diag[1])
    assert len(shared) == 2
    a, b = (Vc.mesh.coords[v] for v in shared)
    left_diagonal = bool(np.allclose(sorted([tuple(a), tuple(b)]), [(0.0, 2.0), (2.0, 0.0)]))
    right_diagonal = bool(np.allclose(sorted([tuple(a), tuple(b)]), [(0.0, 0.0), (2.0, 2.0)]))
    assert left_diagonal != right_diagonal
    # BY HAND: the only vertex of the fine mesh in the interior of the domain is the centre (1, 1).  Its star = the interior of
    # the union of the 6 fine triangles around it (red refinement of two triangles sharing a diagonal: the centre is the midpoint
    # of the diagonal, 3 children of each coarse cell touch it) = the centre itself + the midpoints of the 6 fine edges at it.
    # All 8 other vertices lie on the boundary: their stars hold no interior dof except edge midpoints of interior edges.
    L = lv[1]
    pp, pd = np.asarray(L.patch_ptr), np.asarray(L.patch_dofs)
    centre = int(np.flatnonzero(np.all(np.isclose(Vf.node_coords, [1.0, 1.0]), axis=1))[0])
    stars = [set(pd[pp[i]:pp[i + 1]] // 2) for i in range(len(pp) - 1)]
    big = [s for s in stars if centre in s]
    assert len(big) == 1
    # nodes at distance 1/2 from the centre along the 6 fine edges through it: 4 axis neighbours + 2 on the diagonal
    off = [(0.5, 0), (-0.5, 0), (0, 0.5), (0, -0.5)] + ([(0.5, -0.5), (-0.5, 0.5)] if left_diagonal else [(0.5, 0.5), (-0.5, -0.5)])
    expect = {centre}
    for o in off:
        expect.add(int(np.flatnonzero(np.all(np.isclose(Vf.node_coords, [1.0 + o[0], 1.0 + o[1]]), axis=1))[0]))
    assert big[0] == expect, (big[0], expect)
    # every Dirichlet (boundary) node is in no patch; 16 boundary nodes on the 5 x 5 lattice
    bnd = {i for i, p in enumerate(Vf.node_coords) if min(p) < 1e-12 or max(p) > 2 - 1e-12}
    assert len(bnd) == 16 and not (set().union(*stars) & bnd)
    # the interior dofs number 25 - 16 = 9 nodes; every one of them is in some star
    assert set().union(*stars) == set(range(25)) - bnd
    # BY HAND: interior block of a coarse cell (transfer.py:13-46) = fine nodes strictly inside it = the 3 midpoints of the
    # edges of its central child (each coarse triangle of side 2 has its 4 children; nodes not on the coarse triangle's edges
    # are exactly those 3) -> m = 3 nodes x 2 components = 6 dofs, two blocks
    T = tr[0]
    bd = np.asarray(T.blk_dofs).reshape(2, -1)
    assert bd.shape == (2, 6)
    for blk, cell in zip(bd, Vc.mesh.cells):
        tri = Vc.mesh.coords[cell]
        nodes = sorted(set(blk // 2))
        assert len(nodes) == 3
        for nd in nodes:
            # barycentric coordinates w.r.t. the coarse triangle: all strictly positive, and (by hand) a permutation of
            # (1/2, 1/4, 1/4)
            M = np.concatenate([np.ones((3, 1)), tri], axis=1)
            lam = np.array([1.0, *Vf.node_coords[nd]]) @ np.linalg.inv(M)
            assert np.allclose(sorted(lam), [0.25, 0.25, 0.5])


def test_node_numbering_is_the_z_order_of_the_positions_and_interior_stars_hold_seven_nodes():
    """N = 2 (8 coarse triangles, 32 fine ones, fine vertices on a 5 x 5 lattice, P2 nodes on the 9 x 9 lattice of spacing 1/4):
    (a) the node numbers follow the Z-order (Morton) curve of the positions -- the key recomputed HERE by interleaving the bits
    of the positions quantised to 20 bits; (b) BY HAND: a fine vertex in the interior of the domain has 6 triangles around it, so its star
    holds the vertex and the 6 edge midpoints = 7 nodes = 14 dofs -- the patch size BASELINE.md quotes for configs 1-2 --, there
    are 3 x 3 = 9 such vertices, and the 16 boundary vertices give smaller patches (those with at least one interior edge at
    them) or none."""
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, build_hierarchy
    lv, tr = build_hierarchy(TwoDimLidDrivenCavityProblem(2), 1, 2, Re=0.0, gamma=1.0)
    V = lv[1].V
    assert V.mesh.num_cells == 32 and V.mesh.num_vertices == 25 and V.num_nodes == 81
    ij = np.round(V.node_coords * 4).astype(int)                       # lattice coordinates 0 .. 8
    assert {tuple(p) for p in ij} == {(i, j) for i in range(9) for j in range(9)}

    def z_key(i, j):
        # positions quantised to 20 bits over the extent of the domain (0 .. 8 lattice steps -> 0 .. 2^20 - 1), x in the even bits
        qi, qj = int(i / 8 * (2 ** 20 - 1) + 0.5), int(j / 8 * (2 ** 20 - 1) + 0.5)
        k = 0
        for bit in range(20):
            k |= ((qi >> bit) & 1) << (2 * bit) | ((qj >> bit) & 1) << (2 * bit + 1)
        return k
    keys = [z_key(int(i), int(j)) for i, j in ij]
    assert keys == sorted(keys) and len(set(keys)) == 81
    L = lv[1]
    sizes = np.diff(np.asarray(L.patch_ptr))
    assert int(np.count_nonzero(sizes == 14)) == 9 and sizes.max() == 14
    # every 14-dof patch is centred at an interior vertex (even lattice coordinates strictly inside) and holds the 6 nodes at
    # lattice distance 1 along the fine edges through it
    pd, pp = np.asarray(L.patch_dofs), np.asarray(L.patch_ptr)
    centres = set()
    for p in np.flatnonzero(sizes == 14):
        nodes = sorted(set(pd[pp[p]:pp[p + 1]] // 2))
        c = [n for n in nodes if all(x % 2 == 0 for x in ij[n])]
        assert len(c) == 1
        ci = ij[c[0]]
        assert 0 < ci[0] < 8 and 0 < ci[1] < 8
        centres.add(tuple(ci))
        others = sorted(tuple(ij[n] - ci) for n in nodes if n != c[0])
        assert len(others) == 6 and set(others) >= {(1, 0), (-1, 0), (0, 1), (0, -1)}
        assert set(others) - {(1, 0), (-1, 0), (0, 1), (0, -1)} in ({(1, -1), (-1, 1)}, {(1, 1), (-1, -1)})
    assert centres == {(i, j) for i in (2, 4, 6) for j in (2, 4, 6)}
