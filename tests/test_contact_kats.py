"""Known-answer tests of the contact model that replaces PhysX -- the only pins the contact
physics can have, Isaac Gym being absent: Coulomb stick / slip on an incline, sliding deceleration, impact without
rebound, resting penetration, the regularisation creep, and the joint-limit spring.  Parameters replaced:
reference shifu/configs/env_config.py:50-58 (PhysX contact_offset / bounce_threshold / max_depenetration_velocity)
and :82-84 (terrain friction 1.0, restitution 0); friction combine = average ([EXT] PhysX default).

Each scenario runs on the double AND the float build of the CPU oracle; tests/test_gpu_parity.py::test_contact_kats_*
runs the same scenarios on the HIP kernels (bit-equal to the float oracle, and against the same analytic numbers)."""
import numpy as np
import pytest

from tests import kat_models as K
from tests.helpers import sim_params

PREC = [pytest.param(True, id="f64"), pytest.param(False, id="f32")]


def run_block(oracle, f64, theta, mu_shape, steps, lin=(0, 0, 0), dt_sim=K.DT):
    """Block resting on the plane under gravity tilted by `theta` about y (an incline of that slope)."""
    dt = np.float64 if f64 else np.float32
    cm = K.block_model()
    sp = sim_params(dt=dt_sim, gravity=(K.G * np.sin(theta), 0.0, -K.G * np.cos(theta)))
    pen = 2.0 * K.G * np.cos(theta) / (4 * K.K_N)                 # static sag of the four bottom corners
    root = K.root_row((0, 0, 0.05 - pen), lin=lin, dtype=dt)
    dof = np.zeros((0, 2), dt)
    fr = np.full(1, mu_shape, np.float32)
    traj = []
    for k in range(steps):
        oracle.step(cm.blob, sp, 1, dof, root, friction=fr, f64=f64)
        traj.append(root[0].copy())
    return np.array(traj)


@pytest.mark.parametrize("f64", PREC)
def test_block_sticks_below_the_friction_angle_and_creeps_at_the_regularisation_speed(oracle, f64):
    """tan(theta) = 0.3 < mu = 0.6: the block does not slide.  Regularised Coulomb friction holds a tangential load
    T with a steady creep v = v_eps * T / (mu N) = v_eps * tan(theta) / mu -- 0.01 m/s here, the stated bound on
    'static' creep being v_eps = 0.02 m/s (reached only at the friction limit)."""
    mu_shape, mu = 0.2, 0.6
    th = np.arctan(0.3)
    tr = run_block(oracle, f64, th, mu_shape, 600)
    v = tr[-1, 7]
    assert abs(v - K.V_EPS * 0.3 / mu) < 2e-4, v
    assert v < K.V_EPS
    assert np.abs(tr[-200:, 7] - v).max() < 1e-5                   # steady, no stick-slip chatter
    assert np.abs(tr[-1, 10:13]).max() < 1e-4 and abs(tr[-1, 8]) < 1e-6   # no rotation, no sideways drift
    assert abs(tr[-1, 2] - tr[0, 2]) < 2e-4                        # stays on the surface


@pytest.mark.parametrize("f64", PREC)
def test_block_slides_above_the_friction_angle_with_coulomb_acceleration(oracle, f64):
    """tan(theta) = 1 > mu = 0.6: a = g (sin(theta) - mu cos(theta)) along the slope.  The friction force of a sliding
    point is mu f_n |v_end| / |v_start| (isotropic linearly-implicit law: unconditionally stable, no stick-slip
    chatter on light links), i.e. mu f_n up to a relative error a dt / v = 1 / (step index) when starting from rest:
    the acceleration is right once the block moves (within 1.2 % over the second half-second), and the start-up costs
    a fixed velocity offset of order mu g cos(theta) dt ln(n) (0.17 m/s at dt = 5 ms) that halves with dt."""
    mu_shape, mu = 0.2, 0.6
    th = np.arctan(1.0)
    a = K.G * (np.sin(th) - mu * np.cos(th))
    off = []
    for dt_sim in (K.DT, K.DT / 2):
        n = int(round(1.0 / dt_sim))
        tr = run_block(oracle, f64, th, mu_shape, n, dt_sim=dt_sim)
        assert np.abs(tr[-1, 10:13]).max() < 1e-3                  # slides flat, does not tumble
        acc = (tr[-1, 7] - tr[n // 2 - 1, 7]) / 0.5
        assert abs(acc - a) < 0.012 * a, (acc, a)
        off.append(a * 1.0 - tr[-1, 7])
    assert 0.0 < off[0] < 0.2 and 0.45 < off[1] / off[0] < 0.65, off   # start-up offset: bounded, first order in dt


@pytest.mark.parametrize("f64", PREC)
def test_sliding_block_decelerates_at_mu_g_and_stops(oracle, f64):
    """Level ground, v0 = 1 m/s: deceleration mu g (to the same first order in dt), stopping distance
    v0^2 / (2 mu g), then rest -- no creep without load."""
    mu_shape, mu, v0 = 0.6, 0.8, 1.0
    tr = run_block(oracle, f64, 0.0, mu_shape, 200, lin=(v0, 0, 0))
    tr2 = run_block(oracle, f64, 0.0, mu_shape, 400, lin=(v0, 0, 0), dt_sim=K.DT / 2)
    dec = (v0 - tr[9, 7]) / (10 * K.DT)
    dec2 = (v0 - tr2[19, 7]) / (10 * K.DT)
    assert 0.0 < mu * K.G - dec < 0.06 * mu * K.G, dec
    assert 0.4 < (mu * K.G - dec2) / (mu * K.G - dec) < 0.65       # first order in dt
    d = v0 * v0 / (2 * mu * K.G)
    assert d < tr[-1, 0] < 1.08 * d, (tr[-1, 0], d)
    assert abs(tr[-1, 7]) < 1e-4 and abs(tr2[-1, 7]) < 1e-4        # at rest


@pytest.mark.parametrize("f64", PREC)
def test_dropped_sphere_does_not_rebound_and_rests_at_mg_over_k(oracle, f64):
    """restitution 0 (env_config.py:84).  A 1 kg sphere dropped from 0.5 m (3.1 m/s at impact): the speculative
    contact (physx.contact_offset, env_config.py:54) catches it in the step that would cross the surface, it
    penetrates < 3 mm, never leaves the ground again (depenetration at < 5 % of the impact speed, PhysX caps this with
    max_depenetration_velocity) and rests m g / k = 0.196 mm deep."""
    dt = np.float64 if f64 else np.float32
    cm = K.ball_model()
    sp = sim_params()
    root = K.root_row((0, 0, 0.55), dtype=dt)
    dof = np.zeros((0, 2), dt)
    fr = np.ones(1, np.float32)
    z, vz = [], []
    for k in range(400):
        oracle.step(cm.blob, sp, 1, dof, root, friction=fr, f64=f64)
        z.append(root[0, 2] - 0.05); vz.append(root[0, 9])
    z, vz = np.array(z), np.array(vz)
    hit = int(np.argmax(z < 0))
    assert 3.0 < -vz[:hit + 1].min() < 3.2                         # sqrt(2 g 0.5) = 3.13 m/s
    assert vz[hit:].max() < 0.05 * 3.13, vz[hit:].max()            # outward speed while the spring relaxes
    assert z[hit:].max() < 0.0, z[hit:].max()                      # never airborne again: no bounce
    assert z.min() > -0.003, z.min()                               # bounded penetration
    assert abs(-z[-1] - 1.0 * K.G / K.K_N) < 0.05 * K.G / K.K_N     # static sag
    assert abs(vz[-1]) < 1e-5


@pytest.mark.parametrize("f64", PREC)
def test_joint_limit_spring_holds_the_effort_limit(oracle, f64):
    """A hinge driven into its upper limit with its full URDF effort (30 N m; the A1's calf has 55 N m against the
    same spring): the implicit limit spring k = 2000 N m/rad, d = 20 stops it tau / k = 15 mrad past the limit,
    without chatter."""
    dt = np.float64 if f64 else np.float32
    cm = K.limit_model()
    sp = sim_params()
    dof = np.zeros((1, 2), dt)
    root = K.root_row((0, 0, 1.0), dtype=dt)
    tau = np.full(1, 30.0, dt)
    q = []
    for k in range(600):
        oracle.step(cm.blob, sp, 1, dof, root, effort=tau, f64=f64)
        q.append(dof[0, 0])
    q = np.array(q)
    assert abs(q[-1] - (0.5 + 30.0 / 2000.0)) < 2e-4, q[-1]
    assert abs(dof[0, 1]) < 1e-4
    assert q.max() < 0.5 + 0.08        # it arrives at 24.5 rad/s (30 N m on 0.05 kg m^2 over 0.5 rad): overshoot ~ omega I / (d + dt k)
    tau[:] = 0
    for k in range(400):
        oracle.step(cm.blob, sp, 1, dof, root, effort=tau, f64=f64)
    assert abs(dof[0, 0] - 0.5) < 2e-3 and abs(dof[0, 1]) < 1e-2   # released: back at the limit, at rest


def test_standing_a1_holds_a_horizontal_push_below_the_friction_limit(oracle):
    """A1 standing on stiff position drives (implicit PD, kp 200 / kd 5) with a sustained horizontal push on the trunk:
    40 N (mu m g = 0.9 * 122 N = 110 N) -- the feet hold, the trunk drifts slower than v_eps; 150 N -- it is pushed away."""
    from shifu_amd import _abi
    from tests.helpers import a1_model
    cm = a1_model(default_dof_drive_mode=_abi.DOF_MODE_POS)
    m = cm.blob
    for d in range(m.nd):
        m.kp[d], m.kd[d] = 200.0, 5.0
    q0 = np.array([0.1, 0.8, -1.5, 0.1, 0.8, -1.5, -0.1, 0.8, -1.5, -0.1, 0.8, -1.5], np.float32)
    out = {}
    for F in (40.0, 150.0):
        sp = sim_params()
        dof = np.zeros((12, 2), np.float32); dof[:, 0] = q0
        root = K.root_row((0, 0, 0.32))
        fr = np.full(1, 0.8, np.float32)
        push = np.zeros((m.nb, 3), np.float32)
        for k in range(600):
            push[0, 0] = F if k >= 200 else 0.0
            oracle.step(m, sp, 1, dof, root, pos_target=q0, friction=fr, body_force=push)
        out[F] = root[0].copy()
    assert abs(out[40.0][7]) < K.V_EPS and out[40.0][2] > 0.25 and abs(out[40.0][0]) < 0.05, out[40.0]
    assert out[150.0][0] > 0.5 or out[150.0][2] < 0.15, out[150.0]


# ---- capsule vs box (the ABB rod, shf_boxes.h: segment_box_param): a slider pushes a cube over the ground ----
def _push(oracle, yaw, steps=700, f64=True, x0=0.12, y0=0.0):
    from shifu_amd.abb_task import box_desc
    cm = K.pusher_model(yaw=yaw)
    m = cm.blob
    sp = sim_params()
    dt = np.float64 if f64 else np.float32
    cube = box_desc((0.1, 0.1, 0.1), 0.5, 0.6, False, (x0, y0, 0.05))
    dof = np.zeros((1, 2), dt)
    root = np.zeros((2, 13), dt); root[:, 6] = 1.0
    root[1, :3] = (x0, y0, 0.05 - 0.5 * K.G / (4 * K.K_N))
    vt = np.full(1, 0.05, dt)
    fr = np.ones(1, np.float32)
    hist = []
    for k in range(steps):
        contact, bstate, _ = oracle.scene_step(m, sp, [cube], 1, dof, root, vel_target=vt, friction=fr, f64=f64)
        hist.append((dof[0, 0], dof[0, 1], root[1].copy(), contact.copy()))
    return m, hist


@pytest.mark.parametrize("f64", PREC)
def test_capsule_end_pushes_a_cube_with_the_reaction_newton_asks_for(oracle, f64):
    """End-on (the capsule's axis along the push, its end sphere on the cube's face centre): once in contact the cube moves
    at the slider's speed against its ground friction mu m g (mu = the mean of the two materials, as everywhere), and the
    slider feels exactly that reaction -- the pair law is solved consistently (shf_boxes.h: pair_law; the first version
    treated the cube as a free body on the slider's side and felt 0.41 of it, plus a spurious vertical load).  No lateral
    or vertical force, no spin, penetration F / k, and the velocity drive's droop kd (v* - qd) equals the reaction."""
    m, hist = _push(oracle, np.pi / 2, steps=900, f64=f64, x0=0.25)     # leading end at x = 0.1 + q, cube face at 0.2
    q, qd, cube, contact = hist[-1]
    F = 0.5 * (0.6 + 1.0) * 0.5 * K.G
    f_slider, f_cube = contact[m.nb - 1], contact[m.nb]
    assert abs(cube[7] - qd) < 1e-5 and 0.045 < qd < 0.05
    assert abs(f_slider[0] + F) < 0.01 * F and abs(f_slider[1]) < 1e-3 * F and abs(f_slider[2]) < 1e-2 * F, f_slider
    assert abs(2000.0 * (0.05 - qd) - F) < 0.01 * F                                   # what the drive supplies
    assert abs(f_cube[0]) < 0.01 * F and abs(f_cube[2] - 0.5 * K.G) < 1e-3             # push and friction cancel; weight carried
    assert abs(2.0 * np.arctan2(cube[5], cube[6])) < 1e-6 and abs(cube[1]) < 1e-7
    gap = (cube[0] - 0.05) - (q + 0.1 + 0.02)
    assert -2.5 * F / K.K_N < gap < 0.0, gap
    first = next(k for k, h in enumerate(hist) if h[3][m.nb - 1][0] != 0.0)
    assert first > 100 and all(abs(h[2][7]) < 1e-6 for h in hist[:first])               # untouched until the capsule arrives


def test_capsule_parallel_to_the_face_pushes_with_a_line_contact(oracle):
    """Lying along the face the capsule's distance to the box is flat over the overlap: a line contact, held at both ends
    of that stretch (ShfModel.sph_part; one point at its middle until round 3) and solved for both ends together
    (pair_law_joint); it stays a line contact while the cube is within 5e-4 rad of parallel (the flat-sample tolerance;
    without it the contact hopped between the overlap's ends from step to step and the cube's yaw rate chattered at
    +-0.1 rad/s).  Same clean answers as the end-on push."""
    m, hist = _push(oracle, 0.0, steps=900)
    q, qd, cube, contact = hist[-1]
    F = 0.5 * (0.6 + 1.0) * 0.5 * K.G
    f_slider = contact[m.nb - 1]
    assert abs(cube[7] - qd) < 1e-5 and 0.045 < qd < 0.05
    assert abs(f_slider[0] + F) < 0.01 * F and abs(f_slider[1]) < 1e-3 * F and abs(f_slider[2]) < 1e-2 * F, f_slider
    assert abs(2.0 * np.arctan2(cube[5], cube[6])) < 1e-6 and abs(cube[12]) < 1e-6 and abs(cube[1]) < 1e-7
    wz = np.array([h[2][12] for h in hist[-300:]])
    assert np.abs(wz).max() < 1e-5                                                    # no chatter


def test_line_contact_keeps_an_off_centre_cube_flush(oracle):
    """The cube sits 8 cm to the side: the rod (y in [-0.1, 0.1]) overlaps its face over y in [0.03, 0.10], a stretch that
    still contains the cube's centre line (0.08).  The two ends of the stretch carry unequal shares -- their moments about
    the centre line balance: the end 0.02 away carries 2.5 x what the end 0.05 away does -- and the cube travels flush
    against the rod at the rod's speed without turning.  A single contact at the middle of the stretch pushes 1.5 cm off
    the centre line: it hovered at the flat-sample tolerance with a chattering yaw rate of +-0.04 rad/s ("one
    contact point, no torque"), and so did two points eliminated one by one, which cannot share a load unequally."""
    m, hist = _push(oracle, 0.0, steps=900, y0=0.08)
    q, qd, cube, contact = hist[-1]
    F = 0.5 * (0.6 + 1.0) * 0.5 * K.G
    f_slider = contact[m.nb - 1]
    yaw = 2.0 * np.arctan2(cube[5], cube[6])
    assert abs(cube[7] - qd) < 1e-6 and 0.045 < qd < 0.05
    assert abs(cube[12]) < 1e-6 and abs(cube[8]) < 5e-6, (cube[12], cube[8])          # not turning; sideways creep ~ 1 um/s
    assert 0.0 < yaw < 6e-4                          # a static tilt: the heavier-loaded end sinks F_i / k deeper
    assert abs(cube[1] - 0.08) < 1e-4
    assert abs(f_slider[0] + F) < 0.01 * F and abs(f_slider[1]) < 1e-3 * F and abs(f_slider[2]) < 1e-2 * F, f_slider
    wz = np.array([h[2][12] for h in hist[-400:]])
    assert np.abs(wz).max() < 1e-5                                                    # no chatter


def test_capsule_end_pushes_off_centre_and_turns_the_cube(oracle):
    """Turned by 25 degrees the capsule meets the face with its leading end, off the cube's centre line: the contact
    point is that end (closest-point parameter at an end of the segment) and the cube yaws away from it."""
    m, hist = _push(oracle, np.deg2rad(25.0), steps=500)
    touched = [h for h in hist if h[3][m.nb - 1][0] != 0.0]
    assert len(touched) > 50
    q, qd, cube, contact = hist[-1]
    yaw = 2.0 * np.arctan2(cube[5], cube[6])
    # leading end: a = (+half sin, -half cos) -> y < 0 side pushes the -x face below the centre line: positive z torque
    assert yaw > 0.02 and cube[7] > 0.01, (yaw, cube[7])


# ---- the contact scheme against the ODE it discretises ----
def test_sliding_sphere_converges_to_the_continuous_contact_law(oracle):
    """The linearly-implicit step is a discretisation of a continuous compliant-contact law: normal force
    -k phi - d phi_dot (never negative), friction -mu f_n v_t / max(|v_t|, v_eps) at the contact point.  A sphere set
    sliding at 1 m/s (it decelerates and spins up for 29 ms, then rolls) is integrated with that law by scipy's adaptive
    RK45 to 1e-11 and by the oracle at decreasing dt.  Mid-slide (t = 20 ms) speed, spin and position, and the position
    after the transition (t = 0.4 s), approach the ODE's at first order: every halving of dt below 1.25 ms halves the
    error.  At the shipped dt = 5 ms -- six steps for the whole sliding phase -- the sphere ends 0.6 mm short of the
    ODE's 290 mm with the exact rolling speed 5/7 v0."""
    from scipy.integrate import solve_ivp
    m, r, mu, k, d, veps = 1.0, 0.05, 1.0, K.K_N, K.D_N, K.V_EPS
    inertia = 0.4 * m * r * r
    z0 = r - m * K.G / k

    def rhs(t, s):
        x, z, vx, vz, w = s
        phi = z - r
        fn = max(-k * phi - d * vz, 0.0) if phi < 0 else 0.0
        vt = vx - r * w
        ft = -mu * fn * vt / max(abs(vt), veps)
        return [vx, vz, ft / m, -K.G + fn / m, -r * ft / inertia]
    cm = K.ball_model(r=r, m=m)

    def run(T, dt_sim):
        root = K.root_row((0, 0, z0), lin=(1.0, 0, 0), dtype=np.float64)
        oracle.step(cm.blob, sim_params(dt=dt_sim), 1, np.zeros((0, 2)), root, nsteps=int(round(T / dt_sim)),
                    friction=np.ones(1, np.float32), f64=True)
        return np.array([root[0, 0], root[0, 7], root[0, 11]])                       # x, vx, spin
    for T, cols in ((0.02, [0, 1, 2]), (0.4, [0])):
        ref = solve_ivp(rhs, (0, T), [0.0, z0, 1.0, 0.0, 0.0], rtol=1e-11, atol=1e-13, max_step=1e-5).y[[0, 2, 4], -1]
        e = np.array([np.abs(run(T, h) - ref)[cols] for h in (0.00125, 0.000625, 0.0003125)])
        assert (e[1] < 0.62 * e[0]).all() and (e[2] < 0.62 * e[1]).all(), (T, e)       # first order
        assert (e[2] / np.abs(ref[cols]) < 0.02).all(), (T, e)
    assert abs(ref[1] - 5.0 / 7.0) < 1e-6                                            # the ODE itself: rolling at 5/7 v0
    end = run(0.4, 0.005)
    assert abs(end[0] - ref[0]) < 1e-3 and abs(end[1] - 5.0 / 7.0) < 1e-6 and abs(end[2] * r - end[1]) < 1e-6


def test_implicit_position_drive_converges_to_the_pd_oscillator(oracle):
    """Robot._internal_motor_step's DOF_MODE_POS (reference shifu/units/robot.py:55-64) is an implicit PD drive
    tau = kp (q* - q_end) - kd qd_end.  A hinge (inertia 0.05 kg m^2 about its axis, no gravity) stepped to q* = 0.5 rad
    is the damped oscillator I qdd = kp (q* - q) - kd qd: position and speed against scipy's RK45 converge at first order
    in dt, and at the shipped dt = 5 ms the step response is within 3 % of q* throughout (the implicit drive is stable at
    kp dt^2 / I = 0.05 and stays so at 20 ms, where an explicit PD at these gains would not)."""
    from scipy.integrate import solve_ivp
    kp, kd, inertia, target = 100.0, 5.0, 0.01 + 1.0 * 0.2 * 0.2, 0.5
    cm = K.limit_model(lo=-3.0, up=3.0, effort=1000.0)
    m = cm.blob
    m.drive_mode[0], m.kp[0], m.kd[0] = 1, kp, kd                       # DOF_MODE_POS
    sol = solve_ivp(lambda t, s: [s[1], (kp * (target - s[0]) - kd * s[1]) / inertia], (0, 0.3), [0.0, 0.0], rtol=1e-11,
                    atol=1e-13, dense_output=True)

    def run(T, dt_sim):
        dof = np.zeros((1, 2)); root = K.root_row((0, 0, 1.0), dtype=np.float64)
        oracle.step(m, sim_params(dt=dt_sim), 1, dof, root, nsteps=int(round(T / dt_sim)), pos_target=np.full(1, target), f64=True)
        return dof[0].copy()
    ref = sol.sol(0.05)
    e = np.array([np.abs(run(0.05, h) - ref) for h in (0.00125, 0.000625, 0.0003125)])
    assert (e[1] < 0.6 * e[0]).all() and (e[2] < 0.6 * e[1]).all(), e                  # first order
    for dt_sim in (0.005, 0.02):
        dof = np.zeros((1, 2)); root = K.root_row((0, 0, 1.0), dtype=np.float64)
        worst = 0.0
        for k in range(int(round(0.3 / dt_sim))):
            oracle.step(m, sim_params(dt=dt_sim), 1, dof, root, pos_target=np.full(1, target), f64=True)
            worst = max(worst, abs(dof[0, 0] - sol.sol((k + 1) * dt_sim)[0]))
        assert worst < (0.03 if dt_sim == 0.005 else 0.12) * target, (dt_sim, worst)
        assert abs(dof[0, 0] - target) < 3e-3 and abs(dof[0, 1]) < 5e-2               # settling on the target (overdamped), no ringing
