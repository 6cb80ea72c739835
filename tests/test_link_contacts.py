"""Link contacts (ShfModel.link_collide): the articulation's box-shaped colliders against box actors --
vertex-in-box in both directions, rounded shapes against fixed boxes -- on the CPU oracle: known answers in the simplest
settings (a ram on a rail, a table, a cube), the geometric primitive against brute force, and the dropped-contact counter.
tests/test_gpu_parity.py::test_link_contacts_match_oracle_bitwise holds the HIP kernels to the oracle on the same scenes."""
import numpy as np
import pytest

from tests import kat_models as K
from tests.helpers import sim_params


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


def _box(dim, mass, mu, fixed, pos):
    from shifu_amd.abb_task import box_desc
    return box_desc(dim, mass, mu, fixed, pos)


def _run(oracle, cm, boxes, roots, v_target, steps, f64=True):
    m, sp = cm.blob, sim_params()
    dt = np.float64 if f64 else np.float32
    dof = np.zeros((m.nd, 2), dt)
    root = np.zeros((1 + len(boxes), 13), dt)
    root[:, 6] = 1.0
    for k, p in enumerate(roots):
        root[1 + k, :3] = p
    vt = np.full(m.nd, v_target, dt)
    hist = []
    for _ in range(steps):
        contact, _, _ = oracle.scene_step(m, sp, boxes, 1, dof, root, vel_target=vt, friction=np.ones(1, np.float32), f64=f64)
        hist.append((dof.copy(), root.copy(), contact.copy()))
    return m, hist


def test_link_box_pushes_a_free_cube(oracle):
    """(A) against a free box, pair law: the ram's box (6 cm) meets the face of a 10 cm cube with its four leading
    corners; the cube is carried along at the ram's speed against its ground friction, and the ram feels that reaction
    (four simultaneous pair slots: each eliminates the cube on its own, DESIGN 9.3, hence 10 % not 1 %)."""
    cm = K.box_pusher_model()
    cube = _box((0.1, 0.1, 0.1), 0.5, 0.6, False, (0.2, 0.0, 0.05))
    m, hist = _run(oracle, cm, [cube], [(0.2, 0.0, 0.05 - 0.5 * K.G / (4 * K.K_N))], 0.05, 900)
    dof, root, contact = hist[-1]
    F = 0.5 * (0.6 + 1.0) * 0.5 * K.G
    assert 0.04 < dof[0, 1] < 0.05 and abs(root[1, 7] - dof[0, 1]) < 2e-3, (dof[0, 1], root[1, 7])
    assert abs(contact[m.nb - 1][0] + F) < 0.1 * F, (contact[m.nb - 1], F)
    assert abs(contact[m.nb][2] - 0.5 * K.G) < 0.05                                   # the ground still carries the cube
    gap = (root[1, 0] - 0.05) - (dof[0, 0] + 0.03)                                      # cube face minus ram face
    assert -2e-3 < gap < 1e-4, gap
    first = next(k for k, h in enumerate(hist) if h[2][m.nb - 1][0] != 0.0)
    assert first > 100 and all(abs(h[1][1, 7]) < 1e-6 for h in hist[:first])           # untouched until the ram arrives
    without = K.box_pusher_model()
    without.blob.link_collide = 0
    _, h2 = _run(oracle, without, [cube], [(0.2, 0.0, 0.05 - 0.5 * K.G / (4 * K.K_N))], 0.05, 900)
    assert abs(h2[-1][1][1, 0] - 0.2) < 1e-6, "without link contacts the ram passes through the cube"


def test_link_box_is_stopped_by_a_fixed_table(oracle):
    """(A) against a fixed box, plain contact law: the ram driven down onto a table stalls on its four lower corners with
    the drive's stall force kd v* = 100 N spread over them: penetration F / (4 k)."""
    cm = K.box_pusher_model(centre=(0.0, 0.0, 0.5), axis="0 0 -1")          # (well above the ground plane at z = 0)
    table = _box((0.6, 0.6, 0.1), 0.0, 0.5, True, (0.0, 0.0, 0.3))
    m, hist = _run(oracle, cm, [table], [(0.0, 0.0, 0.3)], 0.05, 1200)
    dof, root, contact = hist[-1]
    F = 2000.0 * 0.05
    pen = dof[0, 0] - (0.2 - 0.05 - 0.03)              # travel beyond first touch (ram bottom at 0.47, table top at 0.35)
    assert abs(dof[0, 1]) < 1e-4, dof
    assert abs(pen - F / (4 * K.K_N)) < 0.15 * F / (4 * K.K_N), (pen, F / (4 * K.K_N))
    assert abs(contact[m.nb - 1][2] - F) < 0.02 * F, contact[m.nb - 1]
    assert np.all(contact[m.nb] == 0.0)                # fixed actors report no net force (as for corner contacts)


def test_free_cube_rests_on_a_link_box(oracle):
    """(B) corners of a free box inside the articulation's box volume: a cube set down on a wide anvil link stays on it --
    its four lower corners carry its weight -- and the anvil's row shows that load."""
    cm = K.box_pusher_model(size=(0.3, 0.3, 0.1), centre=(0.0, 0.0, 0.05), kd=2000.0)
    cube = _box((0.05, 0.05, 0.05), 0.2, 0.6, False, (0.0, 0.0, 0.125))
    m, hist = _run(oracle, cm, [cube], [(0.02, 0.01, 0.1249)], 0.0, 600)
    dof, root, contact = hist[-1]
    # static sag: each of the four slots is eliminated against a quarter of the cube INCLUDING its rotational compliance
    # at the corner (which the other three corners cancel in reality): 0.66 mm instead of m g / 4 k = 0.01 mm
    assert -1.0e-3 < root[1, 2] - (0.1 + 0.025) < 0.0, root[1, :3]
    assert abs(root[1, 9]) < 1e-4 and abs(root[1, 0] - 0.02) < 1e-3
    assert abs(contact[m.nb - 1][2] + 0.2 * K.G) < 0.02 * 0.2 * K.G and abs(contact[m.nb][2] - 0.2 * K.G) < 0.02 * 0.2 * K.G
    # the anvil moves under it: friction carries the cube along (mu g = 7.8 m/s^2 >> the gentle start)
    m, hist = _run(oracle, cm, [cube], [(0.02, 0.01, 0.1249)], 0.03, 800)
    assert abs(hist[-1][1][1, 7] - hist[-1][0][0, 1]) < 2e-3 and hist[-1][1][1, 0] > 0.05


def test_capsule_is_stopped_by_a_fixed_table(oracle):
    """(C) a rounded shape against a FIXED box (against free ones the pair slots always existed): the ram's vertical
    capsule driven down onto the table stalls at F / k."""
    cm = K.box_pusher_model(size=None, axis="0 0 -1", capsule=((0.0, 0.0, 0.5), (0.0, 0.0, 0.4), 0.02))
    table = _box((0.6, 0.6, 0.1), 0.0, 0.5, True, (0.0, 0.0, 0.2))
    m, hist = _run(oracle, cm, [table], [(0.0, 0.0, 0.2)], 0.05, 1200)
    dof, root, contact = hist[-1]
    F = 2000.0 * 0.05
    pen = dof[0, 0] - (0.25 - 0.1 - 0.02)
    assert abs(dof[0, 1]) < 1e-4 and abs(pen - F / K.K_N) < 0.15 * F / K.K_N, (dof, pen)
    assert abs(contact[m.nb - 1][2] - F) < 0.02 * F


def test_more_contacts_than_slots_are_dropped_and_counted(oracle):
    """Three box shapes of one link buried in a big fixed box: 24 corners in contact, 16 slots -- 8 dropped per sub-step,
    counted (the GPU keeps the same count per env in SHF_T_DROPPED)."""
    from shifu_amd import _abi
    more = "".join('<collision><origin xyz="%g 0 0.5"/><geometry><box size="0.04 0.04 0.04"/></geometry></collision>' % x for x in (0.1, -0.1))
    cm = K.box_pusher_model(size=(0.04, 0.04, 0.04), centre=(0.0, 0.0, 0.5), extra_shapes=more)
    big = _box((1.0, 1.0, 0.2), 0.0, 0.5, True, (0.0, 0.0, 0.43))           # top at 0.53: every corner 1 - 5 cm deep
    oracle.dropped(reset=True)
    m, hist = _run(oracle, cm, [big], [(0.0, 0.0, 0.43)], 0.0, 5, f64=False)
    assert oracle.dropped(reset=True) == 5 * (24 - _abi.MAX_LINK_CONTACTS)
    assert np.isfinite(hist[-1][0]).all()


def test_box_primitives_against_brute_force(oracle):
    """The primitives both directions use: vertex in box (inside test, depth = distance to the nearest face, that face's
    normal) from the definition, and sphere vs box (gap to the surface, closest surface point) against a dense sampling of
    the box's surface, on random oriented boxes."""
    rng = np.random.default_rng(0)
    pyoracle = oracle
    for it in range(600):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        c, h = rng.uniform(-1, 1, 3), rng.uniform(0.05, 0.5, 3)
        pt = c + R @ (rng.uniform(-1.3, 1.3, 3) * h)
        inside, phi, n = pyoracle.point_in_box(R, c, h, pt)
        d = R.T @ (pt - c)
        pen = h - np.abs(d)
        assert inside == bool((pen > 0).all())
        if inside:
            ax = int(np.argmin(pen))
            assert abs(phi + pen[ax]) < 1e-12 and np.allclose(n, np.sign(d[ax]) * R[:, ax])
        if it % 6 == 0:
            rad = float(rng.uniform(0.0, 0.05))
            (_, _, _), (phs, ns, rc) = pyoracle.box_primitives(R, c, h, pt, rad)
            # brute force: closest point of the box SURFACE to pt over a fine grid of each face
            g = np.linspace(-1, 1, 81)
            best = None
            for ax in range(3):
                u, v = [k for k in range(3) if k != ax]
                U, V = np.meshgrid(g * h[u], g * h[v], indexing="ij")
                for sgn in (-1.0, 1.0):
                    P = np.zeros(U.shape + (3,)); P[..., ax] = sgn * h[ax]; P[..., u] = U; P[..., v] = V
                    dist = np.linalg.norm(P - d, axis=-1)
                    k = np.unravel_index(np.argmin(dist), dist.shape)
                    if best is None or dist[k] < best[0]:
                        best = (float(dist[k]), P[k])
            gap = (-(best[0]) if inside else best[0]) - rad
            assert abs(phs - gap) < 0.02 * float(h.max()) + 1e-9, (phs, gap, inside)        # grid resolution h / 40
            assert np.linalg.norm(R.T @ (rc - c) - best[1]) < 0.06 * float(h.max()) or inside


def _rot(ax, a):
    ax = np.asarray(ax, float)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx


def _quat(ax, a):
    ax = np.asarray(ax, float)
    return tuple(np.sin(0.5 * a) * ax) + (float(np.cos(0.5 * a)),)


def test_box_box_edge_against_brute_force(oracle):
    """box_box_edge (the separating-axis edge case): on random pairs of oriented boxes, whenever it reports a contact the
    normal is the common perpendicular of one edge of each box, unit, pointing from B to A; the reported gap equals the
    signed distance between those two supporting edges along it; the contact point lies within |gap| of both boxes; and no
    face axis has less overlap (the boxes' extents projected on the 6 face normals overlap by more than on the contact
    normal).  Separated boxes (a face axis clears the offset) never report one."""
    rng = np.random.default_rng(3)
    hits = 0
    for it in range(4000):
        def rq():
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            x, y, z, w = q
            return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                             [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                             [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        RA, RB = rq(), rq()
        hA, hB = rng.uniform(0.03, 0.3, 3), rng.uniform(0.03, 0.3, 3)
        cB = rng.uniform(-0.2, 0.2, 3)
        cA = cB + rng.normal(size=3) * 0.25
        hit, phi, n, r = oracle.box_box_edge(RA, cA, hA, RB, cB, hB, 0.01)

        def overlap(L):      # signed separation of the two boxes' projections on unit axis L (< 0: overlap)
            return abs(L @ (cA - cB)) - (np.abs(RA.T @ L) @ hA + np.abs(RB.T @ L) @ hB)
        faces = [overlap(RA[:, i]) for i in range(3)] + [overlap(RB[:, j]) for j in range(3)]
        if max(faces) >= 0.01:
            assert not hit
            continue
        if not hit:
            continue
        hits += 1
        assert abs(np.linalg.norm(n) - 1.0) < 1e-9 and n @ (cA - cB) > 0
        perp = sorted((abs(n @ RA[:, i]), i) for i in range(3))[0], sorted((abs(n @ RB[:, j]), j) for j in range(3))[0]
        assert perp[0][0] < 1e-6 and perp[1][0] < 1e-6                        # perpendicular to one edge direction of each box
        assert abs(phi - overlap(n)) < 2e-5 and phi < 0.01                      # (the 1e-6 added to |R| in the test inflates the radii slightly)
        assert phi > max(faces)                                                  # the least-overlap axis is this one
        for (Rm, c, h) in ((RA, cA, hA), (RB, cB, hB)):
            d = Rm.T @ (r - c)
            assert np.all(np.abs(d) <= h + abs(phi) + 1e-6)
    assert hits > 150, hits


def test_a_box_rests_crosswise_on_another_boxs_edge(oracle):
    """Edge-edge contact: a fixed box turned 45 degrees about x shows a ridge along x; a free box turned 45
    degrees about y is set down on it ridge to ridge, crosswise.  No vertex of either lies in the other -- the vertex-in-volume
    families see nothing and the box would fall through -- the two edges cross at one point, which carries the weight: the
    box stays, sagging m g / k into the ridge, with the contact force m g on its row.  (Balanced on a point it is an unstable
    equilibrium; the float64 build holds it for the 0.6 s simulated here.)"""
    cm = K.box_pusher_model(centre=(2.0, 0.0, 0.5))                            # an articulation far away (the scene needs one)
    s2 = np.sqrt(2.0)
    ridge = _box((0.4, 0.1, 0.1), 0.0, 0.8, True, (0.0, 0.0, 0.3))
    ridge.quat[:] = _quat([1, 0, 0], np.pi / 4)
    bar = _box((0.1, 0.4, 0.1), 0.5, 0.8, False, (0.0, 0.0, 0.0))
    bar.quat[:] = _quat([0, 1, 0], np.pi / 4)
    m, sp = cm.blob, sim_params()
    top = 0.3 + 0.05 * s2                                                       # the ridge line
    z0 = top + 0.05 * s2 + 0.002                                                 # bar's centre: its lower ridge 2 mm above
    dof = np.zeros((m.nd, 2)); root = np.zeros((3, 13)); root[:, 6] = 1.0
    root[0, :3] = (0, 0, 0)
    root[1, :3] = (0.0, 0.0, 0.3); root[1, 3:7] = ridge.quat[:]
    root[2, :3] = (0.01, 0.0, z0); root[2, 3:7] = bar.quat[:]        # centre of mass over the crossing point
    zs = []
    for it in range(120):
        contact, _, _ = oracle.scene_step(m, sp, [ridge, bar], 1, dof, root, vel_target=np.zeros(m.nd), friction=np.ones(1, np.float32), f64=True)
        zs.append(root[2, 2])
    sag = 0.5 * K.G / K.K_N
    assert abs(root[2, 2] - (top + 0.05 * s2 - sag)) < 3e-5, (root[2, 2], top + 0.05 * s2 - sag)
    assert abs(root[2, 9]) < 1e-4 and np.abs(root[2, 10:13]).max() < 1e-3       # at rest, not rolling off (yet)
    assert abs(contact[m.nb + 1][2] - 0.5 * K.G) < 0.01 * 0.5 * K.G and np.abs(contact[m.nb + 1][:2]).max() < 1e-3
    assert min(zs) > top + 0.05 * s2 - 5 * sag - 1e-4                           # it never dipped through
    # the counterfactual: the same drop with the ridge turned away (flat top 5 cm lower): it falls until the FACE contact
    # (vertex-in-volume) catches it -- the two mechanisms hand over
    flat = _box((0.4, 0.1, 0.1), 0.0, 0.8, True, (0.0, 0.0, 0.3))
    root[1, 3:7] = (0, 0, 0, 1)
    root[2, :3] = (0.01, 0.0, z0); root[2, 7:13] = 0.0
    for it in range(160):
        contact, _, _ = oracle.scene_step(m, sp, [flat, bar], 1, dof, root, vel_target=np.zeros(m.nd), friction=np.ones(1, np.float32), f64=True)
    assert root[2, 2] < z0 - 0.015
