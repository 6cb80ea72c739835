/*
 * shifu_amd.h -- C ABI of the MI355X-native vectorised-env backend.
 *
 * This is the drop-in boundary.  The reference has no
 * FFI: its seam is the Python object returned by `gymapi.acquire_gym()`
 * (reference shifu/gym/isaac_gym.py:213-221) plus `gymtorch.wrap_tensor /
 * unwrap_tensor` (isaac_gym.py:108-134).  Every entry point below names the
 * `gym.*` call(s) it replaces.  Signatures are plain pointers and sizes --
 * no torch types.  Device buffers are allocated by the host language
 * (PyTorch-ROCm in this repo) and *bound* to the sim; the library itself only
 * launches kernels on the stream it is handed and never synchronises.
 *
 * All functions return 0 on success, non-zero on error; shf_last_error() gives
 * the message (Isaac Gym prints and returns None/False; the host wrapper
 * raises instead -- INTEGRATION.md).
 */
#ifndef SHIFU_AMD_H
#define SHIFU_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHF_ABI_VERSION 15

#define SHF_MAX_BODIES 32 /* reported rigid bodies per articulation        */
#define SHF_MAX_DOFS 32
#define SHF_MAX_POINTS 176 /* contact sample points per articulation        */
#define SHF_MAX_BOXES 4   /* extra single-body box actors per env          */
#define SHF_MAX_SPHERES 8 /* collision spheres / capsules (vs boxes) per articulation */
#define SHF_MAX_CAPSULES 16 /* self-collision capsules per articulation      */
#define SHF_MAX_PAIRS 96    /* capsule pairs tested for self-collision       */
#define SHF_MAX_SELF_CONTACTS 8 /* simultaneously active self-contacts per env (further ones are dropped, in pair order) */
#define SHF_MAX_ABOX 16         /* box-shaped collision volumes of the articulation (vs the corners of box actors)   */
#define SHF_MAX_HARD_CONTACTS 16 /* contact constraints one env's velocity-level solve can hold (ShfSimParams.max_contacts <= this);
                                  * the deepest are kept, more are dropped and counted (SHF_T_DROPPED)                 */
#define SHF_MAX_LINK_CONTACTS 16 /* simultaneously active link <-> box-actor contacts per env (further ones are dropped, in
                                  * candidate order, and counted: SHF_T_DROPPED)                                     */

#define SHF_MAX_HULLS 8             /* convex-hull collision shapes (mesh colliders) of an articulation that meet the box actors */
#define SHF_HULL_MAX_VERTS 32       /* per hull, after the model compiler's reduction (PhysX cooks <= 64 [EXT]) */
#define SHF_HULL_MAX_FACES 40       /* polygonal faces (coplanar triangles merged)                               */
#define SHF_HULL_MAX_EDGES 64
#define SHF_HULL_MAX_FACE_VERTS 8   /* vertices of one face's loop                                              */
#define SHF_HULL_MAX_LOOP 160       /* all face loops of a hull together                                        */

/* joint types of a reported body's inboard joint */
enum { SHF_JOINT_ROOT = 0, SHF_JOINT_REVOLUTE = 1, SHF_JOINT_PRISMATIC = 2, SHF_JOINT_WELD = 3 };
/* drive modes: numeric values follow gymapi.DOF_MODE_* (robot.py:55-64) */
enum { SHF_DOF_MODE_NONE = 0, SHF_DOF_MODE_POS = 1, SHF_DOF_MODE_VEL = 2, SHF_DOF_MODE_EFFORT = 3 };

/*
 * Flattened articulation, produced by the model compiler (shifu_amd/model.py)
 * from a URDF.  Replaces gym.load_asset + get_asset_* (units.py:73-89).
 * "Reported" bodies are what rigid_body_state / net_contact_force index
 * (fixed joints collapsed unless dont_collapse, a1.urdf FR_foot_fixed).
 * Welded bodies (kept fixed-joint children) carry no inertia of their own:
 * it is merged into dyn[b], their nearest moving ancestor.
 */
typedef struct ShfModel {
  int32_t nb;         /* reported bodies                                   */
  int32_t nd;         /* degrees of freedom                                */
  int32_t np;         /* contact sample points                             */
  int32_t nlevels;    /* max depth of moving bodies below the root         */
  int32_t fixed_base; /* AssetOptions.fix_base_link                        */
  int32_t gravity_on; /* !AssetOptions.disable_gravity                     */
  int32_t nsph;       /* spheres tested against box actors                 */
  int32_t nklevels;   /* max kinematic depth (welded bodies count)         */

  int32_t parent[SHF_MAX_BODIES]; /* reported parent, -1 for the root      */
  int32_t jtype[SHF_MAX_BODIES];
  int32_t dof[SHF_MAX_BODIES];   /* dof index, -1 if none                  */
  int32_t level[SHF_MAX_BODIES]; /* depth among moving bodies (root = 0)   */
  int32_t dyn[SHF_MAX_BODIES];   /* moving body that carries my inertia    */
  int32_t klevel[SHF_MAX_BODIES]; /* kinematic depth: parent's + 1         */
  int32_t child_start[SHF_MAX_BODIES];
  int32_t child_count[SHF_MAX_BODIES];
  int32_t child_list[SHF_MAX_BODIES]; /* moving children, summed in this order */
  int32_t pt_start[SHF_MAX_BODIES];   /* contact points of a moving body   */
  int32_t pt_count[SHF_MAX_BODIES];

  float tpos[SHF_MAX_BODIES][3]; /* joint origin in parent frame           */
  float trot[SHF_MAX_BODIES][9]; /* row-major rotation parent<-child(q=0)  */
  float axis[SHF_MAX_BODIES][3]; /* joint axis in child frame              */
  float mass[SHF_MAX_BODIES];
  float com[SHF_MAX_BODIES][3];
  float inertia[SHF_MAX_BODIES][6]; /* about com: xx xy xz yy yz zz        */

  float lower[SHF_MAX_DOFS];
  float upper[SHF_MAX_DOFS];
  float vel_limit[SHF_MAX_DOFS];
  float effort[SHF_MAX_DOFS];
  float kp[SHF_MAX_DOFS]; /* dof_props stiffness (robot.py:35-37)          */
  float kd[SHF_MAX_DOFS];
  float armature[SHF_MAX_DOFS];
  float damping[SHF_MAX_DOFS];
  int32_t drive_mode[SHF_MAX_DOFS];
  int32_t dof_body[SHF_MAX_DOFS];

  int32_t pt_body[SHF_MAX_POINTS]; /* reported body the force is logged on */
  float pt_pos[SHF_MAX_POINTS][3]; /* in pt_body's frame                   */
  float pt_radius[SHF_MAX_POINTS];
  /* Evaluation slots for kernels that evaluate the points in rounds of one per lane: slot s < neval evaluates point
   * pt_eval[s], or nothing when pt_eval[s] = -1; pt_slot[i] is the slot of point i.  Results do not depend on the
   * assignment: contacts are folded into their body in point order whatever slot evaluated them.  The model compiler
   * packs the points of each moving body into consecutive slots, in point order, never across a multiple of 32 (empty
   * slots pad the blocks; neval >= np), bodies with the lowest rest pose first: a body lane of the chain-mapped kernel
   * then finds its body's active contacts as one bit field of the round's ballot, and rounds whose points are all far
   * from the ground skip the contact response wave-wide.  shf_sim_set_mapping(SHF_MAP_CHAIN) refuses a model whose
   * pt_eval / pt_slot are not each other's inverse over the occupied slots; the 32-lane chain kernel also needs the
   * packed form.  A model built by hand may use the identity (neval = np) with the other kernels. */
  int32_t pt_eval[SHF_MAX_POINTS];
  int32_t pt_slot[SHF_MAX_POINTS];

  /* Rounded shapes of the articulation tested against the free box actors: a sphere (sph_seg = 0), or a capsule --
   * the segment sph_pos + t sph_seg, t in [0,1], swept by sph_radius (the native form of the URDF <cylinder> under
   * replace_cylinder_with_capsule, asset_config.py:32-46).  A capsule meets a box at the point of its segment
   * closest to the box (exact minimiser of the convex piecewise-quadratic distance), then as a sphere there -- or,
   * when it lies along a face (a stretch of the segment is equally close: line contact), at BOTH ends of that stretch:
   * a capsule is two consecutive records of the same geometry, sph_part = 0 (the closest point, or the first end of the
   * stretch) and sph_part = 1 (the second end; off unless there is a stretch).  Against a free box the two ends are
   * solved together (a joint pair law: one 6x6 elimination of the box for both points).  Spheres: one record, part 0. */
  int32_t sph_body[SHF_MAX_SPHERES];
  float sph_pos[SHF_MAX_SPHERES][3];
  float sph_seg[SHF_MAX_SPHERES][3];
  float sph_radius[SHF_MAX_SPHERES];
  int32_t sph_part[SHF_MAX_SPHERES];

  /* Self-collision (create_actor(..., collision_filter = 0), units.py:68): every collision shape of
   * the URDF as one or two capsules (a sphere is a capsule of zero length; a box is the capsule -- or the two
   * side-by-side capsules -- inscribed along its longest axis), and the pairs PhysX would test: all but shapes of
   * one rigid body and of bodies joined by a joint.  A pair closer than the contact offset responds with the same
   * contact law as a ground contact, equal and opposite on the two bodies.  self_collide = 0 switches it off. */
  int32_t self_collide;
  int32_t ncap;
  int32_t npair;
  int32_t neval;   /* evaluation slots in use (pt_eval above); 0 reads as np */
  int32_t cap_body[SHF_MAX_CAPSULES];
  float cap_a[SHF_MAX_CAPSULES][3];  /* segment end points in cap_body's frame */
  float cap_b[SHF_MAX_CAPSULES][3];
  float cap_radius[SHF_MAX_CAPSULES];
  uint8_t pair_a[SHF_MAX_PAIRS];     /* capsule indices, pair_a < pair_b */
  uint8_t pair_b[SHF_MAX_PAIRS];

  /* Link contacts (create_actor(..., group = env, filter = 0), units.py:68: every shape of an env
   * collides with every other actor's): with link_collide != 0 the articulation's collision shapes meet the box actors
   * beyond the rounded sph_* shapes against free boxes --
   *   (A) every contact sample point pt_* (box / hull vertices: radius 0; spheres, capsule ends: their radius) against
   *       every box actor, free or fixed, as a sphere against a box;
   *   (C) the rounded sph_* shapes against the FIXED boxes (against free ones they always are tested);
   *   (B) the eight corners of every box actor against the articulation's box volumes abox_* (vertex in box);
   *   (E) every box volume abox_* against every box actor, edge across edge: the separating-axis test over the 6 face
   *       normals and the 9 edge cross products; when the axis of least overlap is a cross product (by a clear margin:
   *       faces are preferred; and never when an axis of one box lies within 10 degrees of an axis of the other) the two
   *       boxes touch where one edge of each cross -- one contact there.
   * Vertex-face manifolds in both directions plus the edge-edge point; an edge lying flat on a face (no vertex of either
   * box inside the other) is still not detected.  The same edge-edge test runs between every free box actor and every
   * fixed one.  Against a free box the contact
   * takes the consistent pair law (the box is eliminated exactly), against a fixed one the ground-contact law.  At most
   * SHF_MAX_LINK_CONTACTS are active per env and sub-step, in candidate order -- (body, box actor) pair by pair, body ascending
   * then box ascending; within a pair A (points ascending), C (shapes ascending), B (volume ascending, corner
   * ascending), E (volume ascending) -- more are dropped and counted.  abox: centre and orientation in abox_body's frame. */
  int32_t link_collide;
  int32_t nabox;
  int32_t bounds_ok; /* SHF_BOUNDS_MAGIC once shf_model_bounds() has filled bbox; anything else: the kernels test every candidate */
  int32_t nhull;     /* convex hulls of this articulation in the sim's ShfHullSet (shf_sim_set_hulls; 0: none) -- family (H) below */
  int32_t abox_body[SHF_MAX_ABOX];
  float abox_pos[SHF_MAX_ABOX][3];
  float abox_rot[SHF_MAX_ABOX][9]; /* row-major rotation body <- box */
  float abox_half[SHF_MAX_ABOX][3];
  /* Derived (shf_model_bounds): per reported body, the box (centre[3], half extents[3], body frame, axis-aligned there)
   * around every shape of that body that takes part in link contacts -- its sample points and rounded sph_* shapes grown
   * by their radii, its abox_* volumes; half[0] < 0: the body has none.  The link-contact broad phase: a (body, box actor)
   * pair whose two oriented boxes are separated along one of their six face normals by more than the contact offset
   * (+ 1 cm for rounding) cannot produce an active candidate of any family, so its candidates are never evaluated --
   * results are those of testing every candidate, as the oracle does. */
  float bbox[SHF_MAX_BODIES][6];
  /* Derived too: the sample points and box volumes grouped by reported body, each group ascending, so that a kernel finds
   * the candidates of one (body, box actor) pair as a range -- lc_pt[lc_range[b][0] .. + lc_range[b][1]) are body b's
   * points, lc_abox[lc_range[b][2] .. + lc_range[b][3]) its volumes. */
  int16_t lc_range[SHF_MAX_BODIES][4];
  int16_t lc_pt[SHF_MAX_POINTS];
  int16_t lc_abox[SHF_MAX_ABOX];
} ShfModel;
#define SHF_BOUNDS_MAGIC 0x42534831 /* "BSH1" */

/* Convex-hull collision shapes of the articulation (the reference's links collide through <mesh> colliders,
 * asset/urdf/abb_rod_description/urdf/abb_rod_isaac.urdf:38-113, which PhysX cooks into convex hulls [EXT]; every shape of an
 * env collides with every other: create_actor(..., group, 0), shifu/units/units.py:68).  The model compiler
 * (shifu_amd/model.py: reduce_hull) reduces a mesh to a convex polytope within the limits above: vertices, outward face
 * planes n.x <= d with their vertex loops (counter-clockwise seen from outside), and the edges with the two faces that meet
 * in each.  With ShfModel.link_collide and nhull > 0 every hull meets every box actor -- family (H) of the link contacts,
 * after A / C / B / E of its (body, box actor) pair, hulls ascending: the separating-axis test over the hull's face normals,
 * the box's face normals and the edge pairs that span a face of the Minkowski difference (D. Gregorius, "The separating axis
 * test between convex polyhedra", GDC 2013); the axis of largest separation decides -- an edge pair (by a clear margin) gives
 * ONE contact where the two edges cross; a face gives the clipped face manifold: the other body's most anti-parallel face
 * clipped against the side planes of the reference face, the vertices of what is left within the contact offset of the
 * reference plane, reduced to at most four (the deepest, the farthest from it, the two that span the largest area).
 * Kept outside ShfModel (which the kernels stage in LDS): a device copy is bound as SHF_T_HULLS. */
typedef struct ShfHull {
  int32_t body;                 /* reported body the hull is fixed to */
  int32_t nv, nf, ne;
  float centroid[3];            /* a point strictly inside (body frame): orients the edge axes */
  float pad0;
  float vert[SHF_HULL_MAX_VERTS][3];   /* body frame */
  float plane[SHF_HULL_MAX_FACES][4];  /* outward unit normal, offset */
  uint8_t face_start[SHF_HULL_MAX_FACES], face_count[SHF_HULL_MAX_FACES];   /* the face's loop in face_loop */
  uint8_t face_loop[SHF_HULL_MAX_LOOP];
  uint8_t edge[SHF_HULL_MAX_EDGES][4]; /* vertex, vertex, face, face */
} ShfHull;
typedef struct ShfHullSet {
  int32_t nhull;
  int32_t pad[3];
  ShfHull hull[SHF_MAX_HULLS];
} ShfHullSet;

/* A single-body box actor (gym.create_box, object.py:28-39). */
typedef struct ShfBoxDesc {
  float dim[3];
  float mass; /* 0 or fixed!=0 -> static                                  */
  float friction;
  int32_t fixed;
  float pos[3]; /* default pose (reset value)                             */
  float quat[4];
} ShfBoxDesc;

/* The box actors of one env, in actor order after the articulation (device copy bound as
 * SHF_T_SCENE). */
typedef struct ShfScene {
  int32_t nboxes;
  int32_t flags;   /* SHF_SCENE_FACE_MANIFOLD: a free box and a fixed box (and, with link contacts, a box volume of the articulation
                    * and a box actor) that touch without any vertex or edge-crossing contact -- an edge or a face lying flat on a
                    * face, no corner of either inside the other -- get the clipped face manifold of the same narrow phase as the
                    * hulls (<= 4 points, in the slots corners 0..3 would have).  0 (default): rounds 1-5's families only.  The
                    * compile-time-shaped kernels are built for flags == 0; a scene with flags runs on the run-time-shaped ones. */
  int32_t pad[2];
  ShfBoxDesc box[SHF_MAX_BOXES];
} ShfScene;

/*
 * Simulation parameters.  Replaces gymapi.SimParams (+.physx)
 * (shifu/configs/env_config.py:38-58).  Two contact solvers (field `solver`): the
 * velocity-level projected Gauss-Seidel solve the PhysX fields configure, and the
 * linearly-implicit compliant contact of rounds 1-4.
 */
typedef struct ShfSimParams {
  float dt;
  float gravity[3];
  float contact_k;       /* normal stiffness, N/m                          */
  float contact_d;       /* normal damping, N s/m                          */
  float friction_vel;    /* Coulomb regularisation speed, m/s              */
  float limit_k;         /* joint-limit spring, N m/rad                    */
  float limit_d;         /* joint-limit damper                             */
  float angular_damping; /* AssetOptions.angular_damping default 0.5 (root)*/
  float max_ang_vel;     /* AssetOptions.max_angular_velocity default 64   */
  float max_depen_vel;   /* physx.max_depenetration_velocity = 1.0 (env_config.py:57) */
  float contact_offset;  /* physx.contact_offset = 0.01 m (env_config.py:54): a point this close to a surface is
                          * tested, and responds if it would be below the surface at the end of the step
                          * (speculative contact: an impact is stopped at the surface instead of one step later,
                          * v dt deep)                                                                        */
  /* --- contact solver (ABI v12).  gymapi.SimParams.physx as the reference sets it (shifu/configs/env_config.py:50-58):
   * solver_type = 1, num_position_iterations = 8, num_velocity_iterations = 1, rest_offset = 0,
   * bounce_threshold_velocity = 0.5, max_depenetration_velocity = 1.
   * solver == SHF_SOLVER_COMPLIANT (0; a zero-filled tail reads as this): the linearly-implicit spring-damper law above
   *   (contact_k, contact_d, friction_vel), one articulated-body solve per sub-step -- rounds 1-4's model, an explicit opt-in
   *   since round 5.
   * solver == SHF_SOLVER_PGS (1): rigid unilateral contacts with Coulomb cones at velocity level.  Per sub-step: the free
   *   articulated-body solve, the contact-space response W = J M^-1 J^T of the env's active contacts by impulse propagation
   *   through the articulated-body factors, (pos_iters + vel_iters) projected block Gauss-Seidel sweeps over the contacts in
   *   candidate order (3x3 block solve, cone projection with the normal re-solved when sliding), the impulses applied by one
   *   more inward / outward pass.  Target normal velocity of a contact with gap phi (measured from rest_offset):
   *   phi >= 0: -phi / dt (speculative: it may close the gap, no more); phi < 0: min(erp (-phi) / dt, max_depen_vel);
   *   with restitution > 0 and an approach faster than bounce_threshold: at least -restitution v_n.  contact_k, contact_d,
   *   friction_vel are unused.  At most max_contacts (<= SHF_MAX_HARD_CONTACTS; 0 reads as 8) constraints per env: the
   *   candidates with the smallest gap (ties: candidate order), the others dropped and counted in SHF_T_DROPPED.
   * solver == SHF_SOLVER_TGS (2): the same solve as sub-stepped sweeps -- see the define below. */
  int32_t solver;
  int32_t pos_iters;        /* physx.num_position_iterations                                   */
  int32_t vel_iters;        /* physx.num_velocity_iterations (further sweeps of the same kind)  */
  int32_t max_contacts;
  float rest_offset;        /* physx.rest_offset                                               */
  float bounce_threshold;   /* physx.bounce_threshold_velocity                                 */
  float restitution;        /* shape restitution (Isaac Gym default 0; terrain.restitution = 0, env_config.py:84) */
  float erp;                /* share of a penetration removed per sub-step (Baumgarte); 0 reads as 0.2 */
} ShfSimParams;
#define SHF_SCENE_FACE_MANIFOLD 1
#define SHF_SOLVER_COMPLIANT 0
#define SHF_SOLVER_PGS 1
/* solver == SHF_SOLVER_TGS (2): physx.solver_type = 1, what the reference sets (shifu/configs/env_config.py:50): temporal
 * Gauss-Seidel -- the sweeps of SHF_SOLVER_PGS as pos_iters sub-iterations of length h = dt / pos_iters.  Per sub-iteration the
 * constraint errors are re-evaluated from the advanced relative motion (gap_c += h u_n,c after each sweep: no new collision
 * detection, like PhysX), the targets take the sub-step's horizon (open gap: -gap / h, it may close within this sub-iteration;
 * penetration: min(erp (-gap) / h, max_depen_vel)), and the poses advance with each sub-iteration's velocities: they are
 * integrated with the MEAN of the impulses after the pos_iters sweeps, the velocities with the impulses after the velocity
 * iterations (target of an open gap there: -gap / dt with the gap that is left).  External and drive forces are not sub-stepped:
 * one articulated-body solve per dt, as under SHF_SOLVER_PGS.  (M. Macklin et al., "Small steps in physics simulation", 2019;
 * PhysX 5 "TGS" [EXT].) */
#define SHF_SOLVER_TGS 2

typedef struct ShfTerrain {
  int32_t rows, cols; /* height_samples is (rows, cols) int16, x<->row     */
  float hscale, vscale, border;
  float friction; /* terrain static==dynamic friction (env_config.py:82)   */
  /* 0: height field (gym.add_heightfield, cells split along (i+1,j)-(i,j+1)).
   * 1: the triangle mesh convert_heightfield_to_trimesh builds from the same samples
   *    (gym.add_triangle_mesh, isaac_gym.py:369-385): cells split along (i,j)-(i+1,j+1)
   *    and, where a step is steeper than slope_threshold, vertices shifted by one cell
   *    so that the riser is vertical.  SHF_T_HEIGHTS then carries, after the rows*cols
   *    int16 samples, rows*cols bytes: bits 0-1 dx+1, bits 2-3 dy+1 (shift of the vertex
   *    in cells); bits 4-7: for the cell whose lower corner the vertex is, the rows /
   *    columns of cells a query inside it searches besides the cell itself -- bit 4 row
   *    i-1, bit 5 row i+1, bit 6 column j-1, bit 7 column j+1 (set where a triangle of a
   *    neighbouring cell reaches into this one; none set: that one cell only).           */
  int32_t warped;
  /* Optional hint: a lower bound on the z component of every unit surface normal terrain queries can return (height
   * field: 1 / sqrt(1 + Gx^2 + Gy^2) with Gx, Gy the largest sample-to-sample slopes along x and y).  A sample point
   * whose clearance (z - h) * nz_min exceeds contact_offset + radius cannot be within the contact offset, so its
   * normal need not be evaluated.  0 = unknown: no query is shortened.  A value that is not a true lower bound
   * silently drops contacts -- leave it 0 unless computed from the samples (shifu_amd/backend.py does). */
  float nz_min;
} ShfTerrain;

/* Tensor ids for shf_sim_bind / shf_sim_layout.  State tensors follow the
 * Isaac Gym layouts shifu views (isaac_gym.py:108-134, robot.py:48-53). */
enum {
  SHF_T_DOF_STATE = 0,   /* (N*nd, 2) f32  pos,vel       acquire_dof_state_tensor          */
  SHF_T_ROOT_STATE = 1,  /* (N*A, 13) f32                acquire_actor_root_state_tensor   */
  SHF_T_BODY_STATE = 2,  /* (N*B, 13) f32                acquire_rigid_body_state_tensor   */
  SHF_T_CONTACT = 3,     /* (N*B, 3)  f32                acquire_net_contact_force_tensor  */
  SHF_T_JACOBIAN = 4,    /* (N, nb-1|nb, 6, nd|nd+6) f32 acquire_jacobian_tensor           */
  SHF_T_SIM_DOF = 5,     /* internal copies the solver integrates; refresh_* copies them   */
  SHF_T_SIM_ROOT = 6,    /*   into the user-visible tensors above                          */
  SHF_T_EFFORT = 7,      /* (N*nd) f32   set_dof_actuation_force_tensor                    */
  SHF_T_POS_TARGET = 8,  /* (N*nd) f32   set_dof_position_target_tensor                    */
  SHF_T_VEL_TARGET = 9,  /* (N*nd) f32   set_dof_velocity_target_tensor                    */
  SHF_T_BODY_FORCE = 10, /* (N*B, 3) f32 apply_rigid_body_force_at_pos_tensors (CoM)       */
  SHF_T_FRICTION = 11,   /* (N) f32      per-env shape friction (a1_conditional.py:28-31)  */
  SHF_T_HEIGHTS = 12,    /* (rows*cols) i16 height samples (isaac_gym.py:349-367)          */
  SHF_T_MODEL = 13,      /* sizeof(ShfModel) bytes; shf_sim_bind writes the library's copy into it */
  SHF_T_SIM_CONTACT = 14,/* (N*B, 3) f32 internal net contact force of the last step        */
  SHF_T_SCENE = 15,      /* sizeof(ShfScene) bytes, device copy                            */
  SHF_T_DROPPED = 16,    /* (N) i32: contacts dropped since the host last cleared it -- self-contacts beyond
                          * SHF_MAX_SELF_CONTACTS, link contacts beyond SHF_MAX_LINK_CONTACTS (optional binding)   */
  SHF_T_BODY_FORCE_POS = 17, /* (N*B, 3) f32 apply_rigid_body_force_at_pos_tensors: world points of application */
  SHF_T_BODY_MASS_SCALE = 18, /* (N, nb) f32, optional binding: per-env factor on the mass and rotational inertia of each of the
                          * articulation's bodies (centre of mass unchanged) -- gym.set_actor_rigid_body_properties(env, actor,
                          * props, recomputeInertia=True) with props[b].mass = factor * the asset's (shifu/units/units.py:104-110).
                          * Read once per launch, in the inertia phase; unbound = 1 everywhere.  Bodies welded to their
                          * parent carry no mass of their own here (ShfModel.mass is 0 for them): their factor is unused. */
  SHF_T_HULLS = 19,      /* sizeof(ShfHullSet) bytes, device copy of shf_sim_set_hulls' argument (needed when ShfModel.nhull > 0) */
  SHF_T_CONTACT_HIST = 20, /* (N, SHF_CONTACT_HIST_BINS + 1) i32, optional binding (SHF_SOLVER_PGS): per env, how many sub-steps offered k
                          * candidate constraints to the solve, BEFORE the max_contacts cap -- column k for k < BINS - 1, column BINS - 1
                          * for more; the LAST column is the env's drop counter: while this tensor is bound the kernels count dropped
                          * contacts here and leave SHF_T_DROPPED alone (one more read-modify-write per env and sub-step; unbound: none) */
  SHF_T_COUNT = 21
};
#define SHF_CONTACT_HIST_BINS 26

/* refresh masks: gym.refresh_*_tensor (isaac_gym.py:139-154) */
enum {
  SHF_REFRESH_DOF = 1,
  SHF_REFRESH_ROOT = 2,
  SHF_REFRESH_BODY = 4,
  SHF_REFRESH_CONTACT = 8,
  SHF_REFRESH_JACOBIAN = 16,
  SHF_REFRESH_ALL = 31
};

typedef struct ShfSim ShfSim;

const char* shf_last_error(void);
int shf_abi_version(void);

/* gym.create_sim (isaac_gym.py:217-220) */
int shf_sim_create(const ShfSimParams* params, ShfSim** out);
/* gym.destroy_sim (isaac_gym.py:287) */
int shf_sim_destroy(ShfSim* sim);
/* gym.add_ground (isaac_gym.py:197-203): rows==0 -> flat plane z=0.
 * gym.add_heightfield / add_triangle_mesh (isaac_gym.py:349-385): samples are
 * bound with SHF_T_HEIGHTS. */
int shf_sim_set_terrain(ShfSim* sim, const ShfTerrain* terrain);
/* gym.load_asset + create_actor for the articulated robot (units.py:57-77) */
int shf_sim_set_articulation(ShfSim* sim, const ShfModel* model);
/* Part of gym.load_asset (units.py:73): fills the model's derived broad-phase data (ShfModel.bbox, bounds_ok) from its
 * shape records, in place.  Host only, no GPU.  shf_sim_set_articulation does the same on its own copy, and
 * shf_sim_bind(SHF_T_MODEL, ptr) writes that copy into the bound buffer, so a binding never has to call this; it is
 * exported for tools that inspect the derived data. */
int shf_model_bounds(ShfModel* model);
/* 1 if a scene of this articulation and `nboxes` box actors can run under ShfSimParams.solver = SHF_SOLVER_PGS: an A1-shaped
 * articulation on its own takes the chain-mapped kernels (csrc/shf_chain_hard.h); any other scene the body-per-lane
 * sub-step with the generic solve at 32 lanes per env (csrc/shf_hard.h: at most 32 bodies + box actors, trees of at most 8
 * levels); else 0: create the sim with SHF_SOLVER_COMPLIANT.  Host only.  The gym facade asks this before gym.create_sim's
 * physx settings (env_config.py:50-58) become a ShfSimParams. */
int shf_model_pgs_supported(const ShfModel* model, int32_t nboxes);
/* Part of gym.load_asset for an asset with <mesh> colliders (units.py:73): the articulation's convex hulls (ShfHullSet; call after
 * shf_sim_set_articulation with a model whose nhull equals hulls->nhull, before shf_sim_finalize).  Host copy; the device copy is
 * bound as SHF_T_HULLS.  Refused: counts beyond the SHF_HULL_* limits, a body index outside the model. */
int shf_sim_set_hulls(ShfSim* sim, const ShfHullSet* hulls);
/* The convex narrow phase on its own (tests, tools): for each of n pairs -- pairs_dev (n, 30) = Ra[9] pa[3] hA[3] Rb[9] pb[3] hB[3]:
 * polytope A = the hull *hull_a_dev (device memory) on the pose (Ra, pa), or, when NULL, the box of half extents hA; B the box hB on
 * (Rb, pb) -- the contacts of A with B as the env steps compute them, `lanes` (16 / 32 / 64) lanes sharing the axes of a pair:
 * out_dev (n, 20) = count, normal[3] (from B towards A), then per contact r[3], gap. */
int shf_convex_manifold(int32_t n, const float* pairs_dev, const ShfHull* hull_a_dev_or_null, float offset, int32_t lanes,
                        float* out_dev, void* stream);
/* ShfScene.flags for the scene built by shf_sim_add_box (before shf_sim_finalize). */
int shf_sim_set_scene_flags(ShfSim* sim, int32_t flags);
/* gym.create_box + create_actor (object.py:28-39) */
int shf_sim_add_box(ShfSim* sim, const ShfBoxDesc* box);
/* gym.create_env x N + prepare_sim (isaac_gym.py:94-104).  env_id_offset is
 * the global id of local env 0 (multi-GPU sharding). */
int shf_sim_finalize(ShfSim* sim, int32_t num_envs, int64_t env_id_offset);
/* Shape/dtype the host must allocate for tensor `id`.
 * dtype: 0 f32, 1 i32, 2 i16, 3 u8, 4 i64. */
int shf_sim_layout(const ShfSim* sim, int32_t id, int64_t shape[4], int32_t* ndim, int32_t* dtype);
/* gymtorch.wrap_tensor in reverse: hand the sim a device pointer for `id`. */
int shf_sim_bind(ShfSim* sim, int32_t id, void* device_ptr);
/* Writes every exposed tensor from the default poses (after create_actor the
 * reference sees spawn poses in root_state: units.py:57-70; Q14 fixed). */
int shf_sim_reset_all(ShfSim* sim, const float* default_root_dev /* (A,13) */, const float* default_dof_dev /* (nd) */,
                      const float* env_origins_dev /* (N,3) or NULL */, void* stream);
/* Lanes per env for the kernels (64 = one wavefront per env, default; 32/16 pack
 * 2/4 envs per wavefront).  Not part of the reference API: a tuning knob. */
int shf_sim_set_group(ShfSim* sim, int32_t lanes);
/* Lane mapping of the fused task steps (shf_a1_step, shf_abb_step).  SHF_MAP_BODY: lane = reported body, the tree is
 * walked level by level through LDS hand-offs (any articulation).  SHF_MAP_CHAIN: lane = kinematic chain, the tree's
 * recursions run link after link on the chain's lane.  Two shapes are compiled: a floating root with 4 serial chains of
 * 3 revolute links that end in one welded body -- the Unitree A1, single actor (csrc/shf_chain.h; call before
 * shf_sim_set_group) -- and a fixed base with one chain of 6 revolute links -- the ABB arm in its table / cube / pad
 * scene (csrc/shf_arm.h; shf_abb_step checks the scene).  The A1's chain mapping also runs with self-collision (32 lanes
 * per env) and on the trimesh terrain; the arm's not with link contacts (but see SHF_MAP_CHAIN_SPLIT).  16 or 32 lanes per
 * env.  Same results bit for bit either way.  Not part of the reference API: a tuning knob.  Fails for another shape. */
/* SHF_MAP_CHAIN_SPLIT (the ABB arm at 16 lanes per env only): the chain mapping with the arm and the box actors of an env
 * on different wavefronts of one workgroup, synchronised by workgroup barriers (csrc/shf_api.hip: k_abb_step_ws); with
 * link contacts (ShfModel.link_collide) the box wave evaluates them, 16 envs per 512-thread workgroup. */
enum { SHF_MAP_BODY = 0, SHF_MAP_CHAIN = 1, SHF_MAP_CHAIN_SPLIT = 2 };
int shf_sim_set_mapping(ShfSim* sim, int32_t mapping);

/* gym.simulate (a1_conditional.py:69, robot.py:69, isaac_gym.py:140) */
int shf_sim_step(ShfSim* sim, void* stream);
/* gym.refresh_{dof_state,actor_root_state,rigid_body_state,jacobian,
 * net_contact_force}_tensor(s) (isaac_gym.py:145-154, a1_conditional.py:72) */
int shf_sim_refresh(ShfSim* sim, int32_t mask, void* stream);
/* gym.set_dof_actuation_force_tensor / set_dof_position_target_tensor /
 * set_dof_velocity_target_tensor (robot.py:55-64): copies (N*nd) floats. */
int shf_sim_set_dof_command(ShfSim* sim, int32_t tensor_id, const float* values_dev, void* stream);
/* gym.set_dof_position_target_tensor_indexed (robot.py:78-82); idx are int32
 * actor indices in sim domain (isaac_gym.py:67). */
int shf_sim_set_pos_target_indexed(ShfSim* sim, const float* values_dev, const int32_t* actor_idx_dev,
                                   int32_t n, void* stream);
/* gym.apply_rigid_body_force_at_pos_tensors(force, None) (robot.py:231-236) */
int shf_sim_apply_body_force(ShfSim* sim, const float* force_dev, void* stream);
/* gym.apply_rigid_body_force_at_pos_tensors(force, pos) (robot.py:231-236, LeggedRobot.apply_force_on_base(force, pos)):
 * forces (N*B, 3) at the world points pos_dev (N*B, 3), consumed by the next shf_sim_step: each acts on its body as the
 * force plus the moment of its arm about the body's origin.  pos_dev == NULL: at the centres of mass, as above.  Rows of
 * box actors are ignored, like there. */
int shf_sim_apply_body_force_at_pos(ShfSim* sim, const float* force_dev, const float* pos_dev, void* stream);
/* gym.set_actor_root_state_tensor_indexed (isaac_gym.py:70-73) */
int shf_sim_commit_root_indexed(ShfSim* sim, const float* root_dev, const int32_t* actor_idx_dev, int32_t n,
                                void* stream);
/* gym.set_actor_root_state_tensor (robot.py:99-100) */
int shf_sim_commit_root_all(ShfSim* sim, const float* root_dev, void* stream);
/* gym.set_dof_state_tensor_indexed (robot.py:83-86) */
int shf_sim_commit_dof_indexed(ShfSim* sim, const float* dof_dev, const int32_t* actor_idx_dev, int32_t n,
                               void* stream);
/* 1 when shf_sim_step has the kernel compiled for this sim's arm + scene under its velocity-level solver (the shipped 6-link arm,
 * table / cube / pad, no velocity drives, no self-collision pairs): shf_sim_set_mapping(sim, SHF_MAP_CHAIN_SPLIT) then selects it
 * for gym.simulate (robot.py:69, isaac_gym.py:140) -- identical results, ~0.7 x the time of the run-time-shaped kernel.  With
 * self-collision on (the reference's AbbPushBox: collision filter 0, units.py:68) the answer is 0.  (ABI v15) */
int shf_sim_step_split_supported(const ShfSim* sim);
/* The three indexed commits of one reset as ONE launch (ABI v15): what IsaacGymEnv.reset_idx (isaac_gym.py:54-73) issues for a
 * reset set -- Robot._reset_dof_state's gym.set_dof_position_target_tensor_indexed and gym.set_dof_state_tensor_indexed
 * (robot.py:78-86), then gym.set_actor_root_state_tensor_indexed (isaac_gym.py:70-73) over every actor's rows.  Any part may
 * be empty (n = 0, pointers NULL); the same result as the three calls above in that order. */
int shf_sim_commit_reset(ShfSim* sim, const float* root_dev, const int32_t* root_idx_dev, int32_t n_root, const float* dof_dev,
                         const float* pos_target_dev, const int32_t* dof_actor_idx_dev, int32_t n_dof, void* stream);

/* ------------------------------------------------------------------------
 * Fused A1Conditional env step: everything
 * ShifuVecEnv.step does for examples/a1_conditional in ONE launch per
 * vec-step (env.py:85-106, a1_conditional.py:64-75,116-221,
 * isaac_gym.py:393-433, train.py:12-35), plus one tiny reduction launch for
 * extras["episode"] (env.py:149-158).
 * ---------------------------------------------------------------------- */
typedef struct ShfA1TaskParams {
  int32_t decimation;        /* cfg.control.decimation = 4                  */
  int32_t extra_substep;     /* 1: refresh_state's extra simulate (Q1)      */
  int32_t num_history;       /* 3                                           */
  int32_t num_height_points; /* 187                                         */
  int32_t base_body;         /* rigid_body_dict['base']                     */
  int32_t curriculum;        /* cfg.terrain.curriculum                      */
  int32_t max_terrain_level; /* cfg.terrain.num_rows                        */
  int32_t num_terrain_cols;
  float action_scale;        /* 0.5 (a1_conditional.py:123)                 */
  float clip_actions;        /* 1.0                                         */
  float clip_obs;            /* 100.0                                       */
  float max_episode_length;  /* ceil(episode_length_s / dt) = 500           */
  float max_episode_length_s;
  float env_length;          /* terrain.env_length = 8 m                    */
  float max_push_force;      /* 5 N (a1_conditional.py:83)                  */
  float spawn_xy;            /* 1 m (a1_conditional.py:47)                  */
  float default_pos[3];      /* 0,0,0.42                                    */
  float default_quat[4];
  float default_dof_pos[SHF_MAX_DOFS];
  float p_gain[SHF_MAX_DOFS];
  float d_gain[SHF_MAX_DOFS];
  int32_t num_leg_bodies;    /* bodies whose name has thigh|calf            */
  int32_t leg_bodies[SHF_MAX_BODIES];
  uint64_t seed;             /* counter-based RNG key                       */
} ShfA1TaskParams;

enum {
  SHF_A1_ACTIONS = 0,    /* (N,12) f32  clipped actions (env.actions)              */
  SHF_A1_OBS = 1,        /* (N,259) f32                                            */
  SHF_A1_REW = 2,        /* (N) f32                                                */
  SHF_A1_RESET = 3,      /* (N) u8 (torch.bool)                                    */
  SHF_A1_TIMEOUT = 4,    /* (N) u8                                                 */
  SHF_A1_EP_LEN = 5,     /* (N) i64                                                */
  SHF_A1_COMMAND = 6,    /* (N,3) f32                                              */
  SHF_A1_HISTORY = 7,    /* (N,12,3) f32 HistoryRecorder.history_buf               */
  SHF_A1_REW_SUMS = 8,   /* (6,N) f32 episode_rewards                              */
  SHF_A1_TORQUES = 9,    /* (N,12) f32                                             */
  SHF_A1_BASE_VEL = 10,  /* (N,9) f32 base_lin_vel, base_ang_vel, projected_gravity */
  SHF_A1_HEIGHTS = 11,   /* (N,187) f32 measured_heights                           */
  SHF_A1_HPOINTS = 12,   /* (187,2) f32 height sample grid (base frame)            */
  SHF_A1_PUSH = 13,      /* (N,B,3) f32 rand_force_buf                             */
  SHF_A1_ORIGINS = 14,   /* (N,3) f32 env_origins                                  */
  SHF_A1_LEVELS = 15,    /* (N) i64 terrain_levels                                 */
  SHF_A1_TYPES = 16,     /* (N) i64 terrain_types                                  */
  SHF_A1_TORIGINS = 17,  /* (rows,cols,3) f32 terrain_origins                      */
  SHF_A1_RESET_COUNT = 18, /* (N) i32 per-env episode counter (RNG counter)        */
  SHF_A1_DONE_SUMS = 19, /* (8,N) f32 per-env finished-episode sums (6 terms, level, 1) */
  SHF_A1_STATS = 20,     /* (R+1,16) f32 ring of per-step episode statistics, row = vec-step % R; row R = the latest step:
                          * [0..5] sum of the six reward-term episode sums over the episodes that finished in that
                          * step, [6] sum of terrain levels over all envs, [7] finished count, [8..13] the reference's
                          * extras["episode"] means (sum / count / max_episode_length_s, env.py:149-158; 0 when nothing
                          * finished), [14] mean terrain level, [15] N                                             */
  SHF_A1_PARAMS = 21,    /* sizeof(ShfA1TaskParams) bytes, device copy             */
  SHF_A1_STATS_ACC = 22, /* (R+1,10) i64 exact integer accumulators of the in-kernel reduction (reward sums in 2^-20
                          * fixed point); row R col 0 = vec-steps completed (selects the ring row; part of the state) */
  SHF_A1_COUNT = 23
};

typedef struct ShfA1Task ShfA1Task;
int shf_a1_create(ShfSim* sim, const ShfA1TaskParams* params, ShfA1Task** out);
int shf_a1_destroy(ShfA1Task* task);
int shf_a1_layout(const ShfA1Task* task, int32_t id, int64_t shape[4], int32_t* ndim, int32_t* dtype);
int shf_a1_bind(ShfA1Task* task, int32_t id, void* device_ptr);
/* ShifuVecEnv.step for A1Conditional (env.py:85-106), log_info's reduction (env.py:149-158) included: the row
 * (vec-steps completed so far) % R of SHF_A1_STATS is written by the same launch.  raw_actions: (N,12) policy output.
 * Nothing about the call depends on the step index, so a captured launch can be replayed from a hipGraph. */
int shf_a1_step(ShfA1Task* task, const float* raw_actions_dev, void* stream);
/* The same with the actions of run_policy('random') (shifu/runner/policy_runner.py:38-41: 2 * rand - 1) drawn inside the
 * launch: U(-1, 1) per dof from the task's counter-based generator (Philox4x32-10, seed ShfA1TaskParams.seed, counter =
 * global env id, vec-step index, dof), so that a random-action roll-out is one launch per vec-step and a sharded run
 * draws what the unsharded one does.  The clipped, scaled actions land in SHF_A1_ACTIONS as usual. */
int shf_a1_step_random(ShfA1Task* task, void* stream);
/* ShifuVecEnv.reset_idx(arange(N)) part of reset() (env.py:108-112). */
int shf_a1_reset_all(ShfA1Task* task, void* stream);

/* ------------------------------------------------------------------------
 * Fused AbbPushBox env step (BASELINE config 5): everything ShifuVecEnv.step does for
 * examples/abb_pushbox_vision/a_prior_stage.py in ONE launch per vec-step:
 * AbbRobot.step (:67-73: EE-delta -> workspace clip -> damped-least-squares IK on the EE
 * Jacobian, shifu/units/robot.py:162-182 -> POS targets), decimation x simulate + the
 * refresh_state sub-step, body-state / Jacobian refresh, termination (:104-113), the two
 * rewards (:121-130), on-device reset of arm / cube / goal (:24-58) and the 6-dim
 * observation (:97-102).  The scene must be [arm, table, cube, goal] in actor order.
 * ---------------------------------------------------------------------- */
typedef struct ShfAbbTaskParams {
  int32_t decimation;      /* int(0.1 / 0.02) = 5 (task_config.py:84)                   */
  int32_t extra_substep;   /* refresh_state's simulate (Q1)                             */
  int32_t ee_body;         /* rigid_body_dict['tip0']                                   */
  int32_t cube_actor;      /* 2 */
  int32_t goal_actor;      /* 3 */
  int32_t pad0;
  float clip_actions;      /* 1.0                                                       */
  float clip_obs;          /* 10.0                                                      */
  float max_episode_length;   /* ceil(20 / 0.1) = 200                                   */
  float max_episode_length_s; /* 20                                                     */
  float ee_velocity;       /* 0.2 m/s (task_config.py:61)                               */
  float env_dt;            /* sim.dt * decimation = 0.1 s                               */
  float ik_damping;        /* 0.05 (robot.py:162)                                       */
  float pad1;
  float min_ee_pos[3];     /* workspace box (task_config.py:63-64)                      */
  float max_ee_pos[3];
  float target_quat[4];    /* [0,1,0,0] (a_prior_stage.py:70)                           */
  float default_dof_pos[SHF_MAX_DOFS];
  float actor_default[SHF_MAX_BOXES + 1][7]; /* reset pose of each actor (pos, quat)    */
  float cube_lo[3], cube_hi[3]; /* RandPosBox.pos_range (a_prior_stage.py:30-33)        */
  float goal_lo[3], goal_hi[3];
  uint64_t seed;
} ShfAbbTaskParams;

enum {
  SHF_ABB_ACTIONS = 0,     /* (N,3) f32 clipped actions                                 */
  SHF_ABB_OBS = 1,         /* (N,6) f32                                                 */
  SHF_ABB_REW = 2,         /* (N) f32                                                   */
  SHF_ABB_RESET = 3,       /* (N) u8                                                    */
  SHF_ABB_TIMEOUT = 4,     /* (N) u8                                                    */
  SHF_ABB_SUCCESS = 5,     /* (N) u8  success_buf                                       */
  SHF_ABB_EP_LEN = 6,      /* (N) i64                                                   */
  SHF_ABB_REW_SUMS = 7,    /* (2,N) f32 reward_reaching, reward_success                 */
  SHF_ABB_DOF_TARGETS = 8, /* (N,nd) f32 robot.dof_targets                              */
  SHF_ABB_RESET_COUNT = 9, /* (N) i32                                                   */
  SHF_ABB_DONE_SUMS = 10,  /* (4,N) f32: finished-episode sums of the 2 terms, success, 1 */
  SHF_ABB_STATS = 11,      /* (R+1,8) f32 ring (row R = latest step): [0..3] sums of DONE_SUMS rows, [4,5] episode means / T, [6] success_rate, [7] N */
  SHF_ABB_PARAMS = 12,     /* sizeof(ShfAbbTaskParams) bytes, device copy               */
  SHF_ABB_STATS_ACC = 13,  /* (R+1,10) i64, as SHF_A1_STATS_ACC                          */
  SHF_ABB_COUNT = 14
};

typedef struct ShfAbbTask ShfAbbTask;
int shf_abb_create(ShfSim* sim, const ShfAbbTaskParams* params, ShfAbbTask** out);
int shf_abb_destroy(ShfAbbTask* task);
int shf_abb_layout(const ShfAbbTask* task, int32_t id, int64_t shape[4], int32_t* ndim, int32_t* dtype);
int shf_abb_bind(ShfAbbTask* task, int32_t id, void* device_ptr);
/* ShifuVecEnv.step for AbbPushBox, statistics row included (as shf_a1_step); raw_actions (N,3). */
int shf_abb_step(ShfAbbTask* task, const float* raw_actions_dev, void* stream);
/* run_policy('random') for the ABB task: as shf_a1_step_random (three end-effector action components per env). */
int shf_abb_step_random(ShfAbbTask* task, void* stream);
/* Introspection (resource tables, tests): 1 when shf_abb_step runs this task under SHF_SOLVER_PGS on the 512-thread form of the
 * run-time-shaped kernel (sixteen envs per workgroup: chosen where their LDS fits one CU), else 0. */
int shf_abb_step_pgs_is_wide(const ShfAbbTask* task);
/* reset_idx(arange(N)) (env.py:108-112). */
int shf_abb_reset_all(ShfAbbTask* task, void* stream);

/* ------------------------------------------------------------------------
 * Trainer kernels.  The reference takes its PPO trainer from the un-vendored rsl_rl package
 * (shifu/runner/policy_runner.py:4,52-73): ActorCritic MLPs of nn.Linear + ELU (shifu/configs/policy_config.py:8-16).
 * These three entry points are one layer's forward and backward as MFMA GEMMs (bf16 operands converted on the fly from
 * the fp32 tensors, fp32 accumulation) with bias / ELU / ELU' fused; row-major fp32 tensors, no torch types.
 *   forward          y[M,N]  = act(x[M,K] w[N,K]^T + b[N])                        act: 0 = identity, 1 = ELU
 *   backward_input   dx[M,K] = (dy (.) act'(y))[M,N] w[N,K]                       y = the forward's output, or NULL (identity)
 *   backward_weight  dw[N,K] = (dy (.) act'(y))^T x,  db[N] = column sums          workspace: see *_workspace (floats)
 * Operand precision (process-wide, shf_mlp_set_precision): SHF_MLP_BF16X3 (default) splits every fp32 operand value
 * into a bf16 head and a bf16 tail and accumulates head*head + head*tail + tail*head -- products good to 2^-16
 * relative, fp32-like results, three MFMAs per tile pair (the layers are HBM-bound, so this costs little);
 * SHF_MLP_BF16 rounds operands to bf16 once (2^-9 relative), one MFMA per tile pair.
 * ---------------------------------------------------------------------- */
#define SHF_MLP_BF16 0
#define SHF_MLP_BF16X3 1
/* bf16x3 for forward and input gradient; the weight gradient's operands rounded once: its sum over the batch rows averages
 * the rounding (relative error of dW ~ 2^-9 / sqrt(rows)), and it is half of a layer's time (profiles/r04_train.md). */
#define SHF_MLP_BF16X3_W1 2
int shf_mlp_set_precision(int32_t mode);
int shf_mlp_get_precision(void);
const char* shf_mlp_last_error(void);
int shf_mlp_linear_forward(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t K, int32_t N,
                           int32_t act, void* stream);
int shf_mlp_linear_backward_input(const float* dy, const float* y_or_null, const float* w, float* dx, int32_t M, int32_t K,
                                  int32_t N, void* stream);
/* Row-panel forms of forward / backward_input (round 4; results equal the two calls above bit for bit).  The weights
 * are first laid out in MFMA fragment order as bf16 heads and tails, plain and transposed, by shf_mlp_pack_weights into
 * a caller-owned device buffer of shf_mlp_pack_bytes(K, N) bytes (16-byte aligned) -- once per change of w, i.e. once per
 * optimizer step; the panel kernels then read each activation row once (a block owns whole rows, M is the only grid
 * dimension), take the weight fragments straight from that buffer, and have no barrier in their reduction loop.
 * x / dy / y must be 16-byte aligned, contiguous. */
int shf_mlp_pack_bytes(int32_t K, int32_t N, int64_t* bytes);
int shf_mlp_pack_weights(const float* w, void* pack, int32_t K, int32_t N, void* stream);
int shf_mlp_panel_forward(const float* x, const void* pack, const float* b, float* y, int32_t M, int32_t K, int32_t N,
                          int32_t act, void* stream);
int shf_mlp_panel_backward_input(const float* dy, const float* y_or_null, const void* pack, float* dx, int32_t M, int32_t K,
                                 int32_t N, void* stream);
/* All layers of an MLP forward in ONE launch (a block carries 32 rows of the batch from the input to the output; the
 * activations pass from layer to layer through LDS and go to HBM once, where y[l] is given -- every layer for a training
 * pass, only the last for inference).  dims[0] = input width, dims[l + 1] = width of layer l's output (all <= 512);
 * pack[l] from shf_mlp_pack_weights(w_l, ., dims[l], dims[l + 1]); bias[l] may be null; act[l] as in forward.  Every
 * stored value equals what the chain of shf_mlp_linear_forward calls stores, bit for bit. */
#define SHF_MLP_MAX_CHAIN 6
typedef struct {
  int32_t nlayers;
  int32_t dims[SHF_MLP_MAX_CHAIN + 1];
  const void* pack[SHF_MLP_MAX_CHAIN];
  const float* bias[SHF_MLP_MAX_CHAIN];
  int32_t act[SHF_MLP_MAX_CHAIN];
  float* y[SHF_MLP_MAX_CHAIN];
} ShfMlpChain;
int shf_mlp_chain_forward(const float* x, int32_t M, const ShfMlpChain* chain, void* stream);
/* 1 if these widths (nlayers, dims) fit the kernel's LDS panels at the current precision, else 0 (run the layers one by one). */
int shf_mlp_chain_fits(const ShfMlpChain* chain);
int shf_mlp_backward_weight_workspace(int32_t M, int32_t K, int32_t N, int64_t* floats);
int shf_mlp_linear_backward_weight(const float* dy, const float* y_or_null, const float* x, float* dw, float* db,
                                   float* workspace, int32_t M, int32_t K, int32_t N, void* stream);

/* PPO mini-batch loss with its gradient, one pass (rsl_rl's PPO.update loss block [EXT], which the reference's runner
 * drives: shifu/runner/policy_runner.py:52-73 with PPOConfig.algorithm, shifu/configs/policy_config.py:18-31):
 *   logp_i   = sum_j log N(actions_ij; mu_ij, std_j)          ratio_i = exp(logp_i - old_logp_i)
 *   surr     = mean_i max(-adv_i ratio_i, -adv_i clamp(ratio_i, 1 - clip, 1 + clip))
 *   value    = mean_i max((v_i - R_i)^2, (tv_i + clamp(v_i - tv_i, -clip, clip) - R_i)^2)     (clipped_value != 0)
 *            = mean_i (R_i - v_i)^2                                                          (clipped_value == 0)
 *   entropy  = sum_j (0.5 + 0.5 log 2 pi + log std_j)
 *   kl       = mean_i sum_j log(std_j / old_sigma_ij + 1e-5) + (old_sigma_ij^2 + (old_mu_ij - mu_ij)^2) / (2 std_j^2) - 0.5
 *   loss     = surr + value_coef * value - entropy_coef * entropy
 * out5 = {surr, value, entropy, kl, loss}; dmu[B,A], dstd[A], dvalue[B] = d loss / d (mu, std, value), with torch's
 * conventions at the kinks (max: even split on a tie; clamp: gradient passes on the closed interval).  Row-major fp32,
 * 1 <= A <= 32; sums are taken in a fixed order (no atomics).  workspace: shf_ppo_loss_workspace floats. */
int shf_ppo_loss_workspace(int64_t B, int32_t A, int64_t* floats);
int shf_ppo_loss(const float* mu, const float* std, const float* value, const float* actions, const float* target_values,
                 const float* advantages, const float* returns, const float* old_logp, const float* old_mu,
                 const float* old_sigma, int64_t B, int32_t A, float clip, float value_coef, float entropy_coef,
                 int32_t clipped_value, float* out5, float* dmu, float* dstd, float* dvalue, float* workspace, void* stream);

/* GAE(lambda) of one rollout (rsl_rl RolloutStorage.compute_returns [EXT], driven by shifu/runner/policy_runner.py:52-73
 * with PPOConfig.algorithm.gamma / lam, shifu/configs/policy_config.py:18-31), before the advantage standardisation:
 *   delta_t = r_t + gamma (1 - d_t) V_{t+1} - V_t;  A_t = delta_t + gamma lam (1 - d_t) A_{t+1};  returns_t = A_t + V_t
 * rewards / values / returns (T, N) row-major fp32, dones (T, N) uint8, last_values (N) = V_T.  Same float32 operations in
 * the same order as the torch loop it replaces: identical bits. */
int shf_gae(const float* rewards, const float* values, const unsigned char* dones, const float* last_values, int32_t T,
            int64_t N, float gamma, float lam, float* returns, void* stream);

/* n <= 16 contiguous device buffers copied by one launch (RolloutStorage.add_transitions' per-step writes, rsl_rl [EXT]);
 * src / dst / bytes are HOST arrays of n entries; every buffer 4-byte aligned and a multiple of 4 bytes. */
int shf_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream);
/* rsl_rl OnPolicyRunner.learn's per-step episode bookkeeping [EXT] as one launch:
 *   cur_reward_sum += rewards; cur_episode_length += 1; d = dones != 0;
 *   fin3 += { sum(cur_reward_sum d), sum(cur_episode_length d), sum(d) };  both buffers *= 1 - d
 * rewards / buffers (N) fp32, dones (N) of done_itemsize bytes per entry (1, 2, 4 or 8: bool / integer types), fin3 (3) fp64. */
int shf_episode_bookkeeping(const float* rewards, const void* dones, int32_t done_itemsize, int64_t N,
                            float* cur_reward_sum, float* cur_episode_length, double* fin3, void* stream);

/* RolloutStorage.mini_batch [EXT rsl_rl]: dst_k[i, :] = src_k[idx[i], :] for n <= 16 row-major tensors in one launch; src /
 * dst / row_bytes are HOST arrays, idx_dev (rows) int64 on the device; rows 4-byte aligned multiples of 4 bytes. */
int shf_gather_rows(const void* const* src, void* const* dst, const int32_t* row_bytes, int32_t n, const int64_t* idx_dev,
                    int64_t rows, void* stream);
/* PPO's adaptive learning-rate rule [EXT rsl_rl PPO.update, schedule == 'adaptive'] on device scalars:
 *   lr = kl > kl_high ? max(lr * inv_down, lr_min) : (0 < kl < kl_low ? min(lr * up, lr_max) : lr) */
int shf_adapt_lr(const float* kl_dev, float* lr_dev, float kl_high, float kl_low, float inv_down, float up, float lr_min,
                 float lr_max, void* stream);

/* ---- library glue of the hook-compatible path (ShifuVecEnv with torch hooks), one launch each ----------------------
 * The fused steps carry these inside their kernels; on the source-compatible path they were dozens of small torch
 * launches of LIBRARY code per vec-step.  Same arithmetic as the in-kernel versions and as the oracle's glue_*. */
/* LeggedRobot.post_step (shifu/units/robot.py:222-229): gravity_vec[e] = -1 on up_axis; base_lin_vel / base_ang_vel /
 * projected_gravity = quat_rotate_inverse(root quat, root lin vel / ang vel / gravity_vec), root row = root_idx[e]
 * (or e when NULL) of the (num_root_rows, 13) root-state tensor.  Outputs (n, 3). */
int shf_base_frame_state(const float* root_state, const int64_t* root_idx_or_null, int64_t num_root_rows, int32_t n,
                         int32_t up_axis, float* base_lin_vel, float* base_ang_vel, float* projected_gravity,
                         float* gravity_vec, void* stream);
/* TerrainGymEnv.get_heights (shifu/gym/isaac_gym.py:412-433): yaw-rotate the (num_points, 2) base-frame grid by each
 * env's root quaternion, add root xy and the border, divide by hscale, truncate toward zero, clip to the map, MIN of the
 * cell and its +x / +y neighbours, times vscale.  rows == 0 (plane): zeros.  out (n, num_points). */
int shf_get_heights(const ShfTerrain* terrain, const int16_t* height_samples, const float* root_state,
                    const int64_t* root_idx_or_null, int64_t num_root_rows, const float* height_points_xy, int32_t n,
                    int32_t num_points, float* out, void* stream);
/* HistoryRecorder.add (shifu/utils/train.py:12-14) on a (rows, num_history) view: shift towards the past, x into column 0. */
int shf_history_add(float* history, const float* x, int64_t rows, int32_t num_history, void* stream);
/* buf[idx[i], :] = value for i < n_idx, rows of row_words floats (HistoryRecorder.reset_idx, train.py:16-17; the
 * per-key `sums[env_ids] = 0`); out-of-range indices are skipped. */
int shf_rows_fill_indexed(float* buf, const int64_t* idx, int32_t n_idx, int64_t num_rows, int32_t row_words, float value,
                          void* stream);
/* ShifuVecEnv.log_info (shifu/gym/env.py:149-158) for one reset set: out_means[k] = mean(sums[k][env_ids]) /
 * episode_length_s, then sums[k][env_ids] = 0, for num_keys <= 16 per-env (num_envs) fp32 tensors (`sums` is a HOST array
 * of device pointers); env_ids distinct.  Exact 2^-20 fixed-point sums: order-independent.  workspace17: 17 int64 on the
 * device, zero before the first call (the kernel leaves it zero). */
int shf_episode_log(float* const* sums, int32_t num_keys, const int64_t* env_ids, int32_t n_ids, int64_t num_envs,
                    float episode_length_s, int64_t* workspace17, float* out_means, void* stream);
/* The buffer part of ShifuVecEnv.reset_idx (shifu/gym/env.py:114-130) for one reset set as ONE launch (ABI v15):
 * episode_length[env_ids] = 0, reset_buf[env_ids] = 1 (elements of reset_elem_bytes = 1 or 8 bytes: a bool / uint8 or an
 * int64 tensor), the action history's rows zeroed (HistoryRecorder.reset_idx, train.py:16-17: history_row_words floats per
 * env) and log_info (env.py:149-158) exactly as shf_episode_log does it.  Every optional tensor may be NULL. */
int shf_reset_bookkeeping(float* const* sums, int32_t num_keys, const int64_t* env_ids, int32_t n_ids, int64_t num_envs,
                          float episode_length_s, int64_t* workspace17, float* out_means, int64_t* episode_length_or_null,
                          void* reset_buf_or_null, int32_t reset_elem_bytes, float* history_or_null, int32_t history_row_words,
                          void* stream);
/* Robot._reset_dof_state (shifu/units/robot.py:74-86) ahead of its two indexed commits: for every env id,
 * dof_targets[e, :] = dof_state[e, :, 0] = default_dof_pos, dof_state[e, :, 1] = 0 on the (num_envs, num_dof[, 2])
 * tensors, and actor_ids_out[i] = (int32) root_idx[e] -- the index tensor set_dof_*_tensor_indexed takes. */
int shf_reset_dof_rows(float* dof_state, float* dof_targets, const float* default_dof_pos, const int64_t* env_ids,
                       int32_t n_ids, int64_t num_envs, int32_t num_dof, const int64_t* root_idx_or_null,
                       int32_t* actor_ids_out_or_null, void* stream);
/* ArmRobot.inverse_kinematics (shifu/units/robot.py:162-182): dof_targets_out[e, :] = dof_pos[e, :] + J^T (J J^T +
 * damping^2 I)^-1 dpose, dpose = [goal_pos - ee_pos; orientation_error(goal_quat, ee_quat)] (quat_mul / quat_conjugate of
 * shifu/utils/torch_utils.py:12-40), J = the end effector's (6, num_dof) block at j_ee + e * j_env_stride floats (a view
 * into the Jacobian tensor), ee_pose row (pos, quat xyzw) at ee_pose + e * ee_env_stride, goal_pose (n, 7), dof_pos element
 * (e, d) at dof_pos[(e * num_dof + d) * dof_elem_stride] (2 for the position column of dof_state).  LDL^T, not
 * torch.inverse: the fused ABB step's and the oracle's arithmetic (golden G9: 3e-5 of the row scale against torch). */
int shf_ik_dls(const float* j_ee, int64_t j_env_stride, const float* dof_pos, int32_t dof_elem_stride, const float* ee_pose,
               int64_t ee_env_stride, const float* goal_pose, int32_t n, int32_t num_dof, float damping,
               float* dof_targets_out, void* stream);
/* ShifuVecEnv.compute_reward's accumulation (env.py:180-185): rew = terms[0] + terms[1] + ... in that order,
 * sums[k] += terms[k]; HOST arrays of num_keys <= 16 device pointers to (num_envs) fp32 tensors. */
int shf_reward_accumulate(const float* const* terms, float* const* sums, int32_t num_keys, int64_t num_envs, float* rew,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SHIFU_AMD_H */
