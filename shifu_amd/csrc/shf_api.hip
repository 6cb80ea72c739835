// shf_api.hip -- kernels and the C-ABI entry points of include/shifu_amd.h.
//
// The library owns no device memory: every buffer is bound by the host
// (shf_sim_bind / shf_a1_bind), kernels are launched on the stream handed in
// and nothing here synchronises.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>
#include <algorithm>
#include <cmath>
#include <string>

#include "shf_device.h"
#include "shf_task.h"
#include "shf_arm.h"

// ------------------------------------------------------------ host state --
static thread_local std::string g_err;
static int fail(const std::string& msg) { g_err = msg; return 1; }
int shf_set_error(const std::string& msg) { return fail(msg); }   // for the other translation units (shf_glue.hip)
#define HIP_OK(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct ShfSim {
  ShfSimParams sp;
  ShfTerrain terr;
  ShfModel model;
  bool has_model = false;
  int nboxes = 0;
  ShfBoxDesc boxes[SHF_MAX_BOXES];
  int32_t n = 0;
  int64_t env_off = 0;
  bool finalized = false;
  bool force_armed = false;
  bool force_at_pos = false;   // the armed force comes with points of application (SHF_T_BODY_FORCE_POS)
  void* t[SHF_T_COUNT] = {};
  int group = 64;  // lanes per env
  int mapping = SHF_MAP_BODY;
  int mapping_split = 0;   // SHF_MAP_CHAIN_SPLIT: threads per block of the wave-specialised arm step (256 / 512), else 0
  int chain_group = 16;  // lanes per env of the chain-mapped fused A1 step (the other kernels keep `group`)
  ShfHullSet hulls;      // the articulation's convex hulls (shf_sim_set_hulls), model.nhull of them
  bool has_hulls = false;
  int scene_flags = 0;   // ShfScene.flags
};

struct ShfA1Task {
  ShfSim* sim;
  ShfA1TaskParams tp;
  void* t[SHF_A1_COUNT] = {};
  int stats_ring = 256;
};

// shf_a1_chain.hip: the chain-mapped fused A1 step (its own translation unit)
bool shf_a1_chain_matches(const ShfModel& m);
size_t shf_a1_chain_lds_bytes(int G, int nobs, bool self);
const void* shf_a1_chain_kernel(int G, bool warped, bool self);
const void* shf_a1_chain_pgs_kernel(bool warped, bool self, bool k16, bool tgs);
int shf_a1_chain_pgs_max_contacts(void);
const void* shf_sim_step_chain_pgs_kernel(bool warped, bool self, bool k16, bool tgs);
size_t shf_sim_step_chain_pgs_lds_bytes(bool self);
#ifdef SHF_PHASE_CLOCK
int shf_a1_chain_phase_cycles(unsigned long long* out, int n, int reset);
#endif

extern "C" const char* shf_last_error(void) { return g_err.c_str(); }
extern "C" int shf_abi_version(void) { return SHF_ABI_VERSION; }


// The kernels: templates in shf_kernels.h; their instantiations are compiled by the shf_k_*.hip translation units (one family
// each, side by side) and are extern templates here.  -DSHF_UNITY (debug builds that share one set of device globals, e.g. the
// phase clock) compiles everything in this one translation unit instead.
#define SHF_DEFINE_SMALL_KERNELS
#ifdef SHF_UNITY
#define SHF_DEFINE_A1_KERNELS
#endif
#include "shf_kernels.h"
#ifndef SHF_UNITY
#define SHF_KERNEL(...) extern template __global__ void __VA_ARGS__;
#include "shf_kernel_list.h"
#undef SHF_KERNEL
#endif


// ---------------------------------------------------------------- C ABI --
static bool sim_self(const ShfSim* s) { return s->model.self_collide != 0 && s->model.npair > 0; }
static bool sim_link(const ShfSim* s) { return s->model.link_collide != 0 && s->nboxes > 0; }
// the compile-time-shaped kernels are built for scenes without the face-manifold family and without hulls: such a scene runs on the run-time-shaped ones
static bool sim_plain(const ShfSim* s) { return s->scene_flags == 0 && s->model.nhull == 0; }
static int sim_ndyn(const ShfSim* s) {   // free boxes: the only ones that own contact slots (SlotLay)
  int n = 0;
  for (int k = 0; k < s->nboxes; k++) n += (!s->boxes[k].fixed && s->boxes[k].mass > 0.0f) ? 1 : 0;
  return n;
}
static size_t sim_lds_bytes(const ShfSim* s, int head_words, int min_tail, bool boxes = false) {
  const int epb = 256 / s->group;
  const int nbx = boxes ? s->nboxes : 0;
  int nslots = s->model.np + (boxes ? box_slot_count(nbx, sim_ndyn(s), s->model.nsph) : 0) + (sim_self(s) ? SHF_MAX_SELF_CONTACTS : 0) +
               ((boxes && sim_link(s)) ? 2 * SHF_MAX_LINK_CONTACTS : 0);
  if (s->sp.solver != SHF_SOLVER_COMPLIANT) nslots = hard_total_slots(nslots, boxes && sim_link(s));
  return ((size_t)MODEL_WORDS + head_words + (boxes ? SCENE_WORDS : 0) +
          (size_t)epb * env_lds_words(s->model.nb + nbx, s->model.nd, nslots, min_tail, 1 + (boxes ? nbx : s->nboxes))) * 4;
}

extern "C" int shf_sim_create(const ShfSimParams* params, ShfSim** out) {
  if (!params || !out) return fail("shf_sim_create: null argument");
  if (!(params->dt > 0.0f)) return fail("shf_sim_create: dt must be > 0");
  ShfSim* s = new ShfSim();
  s->sp = *params;
  memset(&s->terr, 0, sizeof s->terr);
  s->terr.friction = 1.0f;
  *out = s;
  return 0;
}
extern "C" int shf_model_pgs_supported(const ShfModel* model, int32_t nboxes) {
  if (!model || nboxes < 0 || nboxes > SHF_MAX_BOXES) return 0;
  if (nboxes == 0 && shf_a1_chain_matches(*model)) return 1;
  return (model->nlevels <= HG_LEV && model->nb + nboxes <= 32) ? 1 : 0;
}
extern "C" int shf_sim_destroy(ShfSim* sim) {
  delete sim;
  return 0;
}
extern "C" int shf_sim_set_terrain(ShfSim* sim, const ShfTerrain* terrain) {
  if (!sim || !terrain) return fail("shf_sim_set_terrain: null argument");
  if (terrain->rows != 0 && (terrain->rows < 2 || terrain->cols < 2 || !(terrain->hscale > 0.0f)))
    return fail("shf_sim_set_terrain: bad heightfield shape/scale");
  if (terrain->warped && terrain->rows == 0) return fail("shf_sim_set_terrain: a warped (trimesh) terrain needs height samples");
  if (terrain->warped && sim->nboxes > 0) return fail("shf_sim_set_terrain: trimesh terrain with box actors is not supported");
  sim->terr = *terrain;
  return 0;
}
extern "C" int shf_sim_set_articulation(ShfSim* sim, const ShfModel* model) {
  if (!sim || !model) return fail("shf_sim_set_articulation: null argument");
  if (model->nb < 1 || model->nb > SHF_MAX_BODIES || model->nd < 0 || model->nd > SHF_MAX_DOFS || model->np < 0 ||
      model->np > SHF_MAX_POINTS)
    return fail("shf_sim_set_articulation: model exceeds SHF_MAX_*");
  sim->model = *model;
  shf_model_bounds(&sim->model);
  sim->has_model = true;
  return 0;
}
// ShfModel.bbox: the bounding box (body frame) of everything a body contributes to link contacts -- sample points grown
// by their radii, the rounded shapes' segments grown by theirs, the eight vertices of each box volume -- rounded outwards
static int model_bounds(ShfModel* m, const ShfHullSet* hs);
extern "C" int shf_model_bounds(ShfModel* m) { return model_bounds(m, nullptr); }
// (hs: the articulation's convex hulls, whose vertices count towards their bodies' boxes)
static int model_bounds(ShfModel* m, const ShfHullSet* hs) {
  if (!m) return fail("shf_model_bounds: null model");
  if (m->nb < 1 || m->nb > SHF_MAX_BODIES || m->np < 0 || m->np > SHF_MAX_POINTS || m->nsph < 0 || m->nsph > SHF_MAX_SPHERES ||
      m->nabox < 0 || m->nabox > SHF_MAX_ABOX)
    return fail("shf_model_bounds: model exceeds SHF_MAX_*");
  struct Ball { double c[3], r; int body; };
  std::vector<Ball> balls;
  for (int i = 0; i < m->np; i++) balls.push_back({{m->pt_pos[i][0], m->pt_pos[i][1], m->pt_pos[i][2]}, m->pt_radius[i], m->pt_body[i]});
  for (int i = 0; i < m->nsph; i++)
    for (int e = 0; e < 2; e++)
      balls.push_back({{m->sph_pos[i][0] + e * m->sph_seg[i][0], m->sph_pos[i][1] + e * m->sph_seg[i][1], m->sph_pos[i][2] + e * m->sph_seg[i][2]},
                       m->sph_radius[i], m->sph_body[i]});
  for (int j = 0; j < m->nabox; j++)
    for (int c = 0; c < 8; c++) {
      const double lc[3] = {((c & 4) ? 1.0 : -1.0) * m->abox_half[j][0], ((c & 2) ? 1.0 : -1.0) * m->abox_half[j][1], ((c & 1) ? 1.0 : -1.0) * m->abox_half[j][2]};
      Ball b = {{0, 0, 0}, 0.0, m->abox_body[j]};
      for (int r = 0; r < 3; r++)
        b.c[r] = m->abox_pos[j][r] + m->abox_rot[j][3 * r] * lc[0] + m->abox_rot[j][3 * r + 1] * lc[1] + m->abox_rot[j][3 * r + 2] * lc[2];
      balls.push_back(b);
    }
  for (int j = 0; hs && j < hs->nhull; j++)
    for (int i = 0; i < hs->hull[j].nv; i++) balls.push_back({{hs->hull[j].vert[i][0], hs->hull[j].vert[i][1], hs->hull[j].vert[i][2]}, 0.0, hs->hull[j].body});
  for (int b = 0; b < SHF_MAX_BODIES; b++) {
    double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
    bool any = false;
    for (const Ball& s : balls)
      if (s.body == b) {
        any = true;
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], s.c[k] - s.r); hi[k] = std::max(hi[k], s.c[k] + s.r); }
      }
    for (int k = 0; k < 6; k++) m->bbox[b][k] = 0.0f;
    if (!any || b >= m->nb) { m->bbox[b][3] = -1.0f; continue; }
    for (int k = 0; k < 3; k++) {
      m->bbox[b][k] = (float)(0.5 * (lo[k] + hi[k]));
      const double reach = std::max(hi[k] - (double)m->bbox[b][k], (double)m->bbox[b][k] - lo[k]);
      m->bbox[b][3 + k] = (float)(reach * (1.0 + 1e-5) + 1e-6);   // rounded up
    }
  }
  // shapes grouped by body (each group ascending): lc_range / lc_pt / lc_abox
  int npt = 0, nab = 0;
  for (int b = 0; b < SHF_MAX_BODIES; b++) {
    m->lc_range[b][0] = (int16_t)npt; m->lc_range[b][2] = (int16_t)nab;
    if (b < m->nb) {
      for (int i = 0; i < m->np; i++) if (m->pt_body[i] == b) m->lc_pt[npt++] = (int16_t)i;
      for (int j = 0; j < m->nabox; j++) if (m->abox_body[j] == b) m->lc_abox[nab++] = (int16_t)j;
    }
    m->lc_range[b][1] = (int16_t)(npt - m->lc_range[b][0]); m->lc_range[b][3] = (int16_t)(nab - m->lc_range[b][2]);
  }
  for (int i = npt; i < SHF_MAX_POINTS; i++) m->lc_pt[i] = 0;
  for (int j = nab; j < SHF_MAX_ABOX; j++) m->lc_abox[j] = 0;
  m->bounds_ok = SHF_BOUNDS_MAGIC;
  return 0;
}
extern "C" int shf_sim_set_hulls(ShfSim* sim, const ShfHullSet* hs) {
  if (!sim || !hs) return fail("shf_sim_set_hulls: null argument");
  if (!sim->has_model) return fail("shf_sim_set_hulls: before shf_sim_set_articulation");
  if (sim->finalized) return fail("shf_sim_set_hulls: after shf_sim_finalize");
  if (hs->nhull < 0 || hs->nhull > SHF_MAX_HULLS || hs->nhull != sim->model.nhull)
    return fail("shf_sim_set_hulls: nhull must equal ShfModel.nhull (<= SHF_MAX_HULLS)");
  for (int j = 0; j < hs->nhull; j++) {
    const ShfHull& h = hs->hull[j];
    if (h.body < 0 || h.body >= sim->model.nb) return fail("shf_sim_set_hulls: hull on a body outside the model");
    if (h.nv < 4 || h.nv > SHF_HULL_MAX_VERTS || h.nf < 4 || h.nf > SHF_HULL_MAX_FACES || h.ne < 6 || h.ne > SHF_HULL_MAX_EDGES)
      return fail("shf_sim_set_hulls: hull exceeds the SHF_HULL_* limits");
    for (int f = 0; f < h.nf; f++) {
      if (h.face_count[f] < 3 || h.face_count[f] > SHF_HULL_MAX_FACE_VERTS || h.face_start[f] + h.face_count[f] > SHF_HULL_MAX_LOOP)
        return fail("shf_sim_set_hulls: bad face loop");
      for (int k = 0; k < h.face_count[f]; k++)
        if (h.face_loop[h.face_start[f] + k] >= h.nv) return fail("shf_sim_set_hulls: face loop names a vertex beyond nv");
    }
    for (int e = 0; e < h.ne; e++)
      if (h.edge[e][0] >= h.nv || h.edge[e][1] >= h.nv || h.edge[e][2] >= h.nf || h.edge[e][3] >= h.nf) return fail("shf_sim_set_hulls: bad edge record");
  }
  sim->hulls = *hs;
  sim->has_hulls = hs->nhull > 0;
  return model_bounds(&sim->model, &sim->hulls);       // the bodies' broad-phase boxes now hold the hulls too
}
extern "C" int shf_sim_set_scene_flags(ShfSim* sim, int32_t flags) {
  if (!sim) return fail("shf_sim_set_scene_flags: null sim");
  if (sim->finalized) return fail("shf_sim_set_scene_flags: after shf_sim_finalize");
  if (flags & ~SHF_SCENE_FACE_MANIFOLD) return fail("shf_sim_set_scene_flags: unknown flag");
  sim->scene_flags = flags;
  return 0;
}
extern "C" int shf_sim_add_box(ShfSim* sim, const ShfBoxDesc* box) {
  if (!sim || !box) return fail("shf_sim_add_box: null argument");
  if (sim->nboxes >= SHF_MAX_BOXES) return fail("shf_sim_add_box: too many boxes");
  if (sim->terr.warped) return fail("shf_sim_add_box: box actors on a trimesh terrain are not supported");
  sim->boxes[sim->nboxes++] = *box;
  return 0;
}
extern "C" int shf_sim_finalize(ShfSim* sim, int32_t num_envs, int64_t env_id_offset) {
  if (!sim || !sim->has_model) return fail("shf_sim_finalize: no articulation set");
  if (num_envs <= 0) return fail("shf_sim_finalize: num_envs must be > 0");
  sim->n = num_envs;
  sim->env_off = env_id_offset;
  sim->finalized = true;
  return 0;
}
extern "C" int shf_sim_set_group(ShfSim* sim, int32_t lanes) {
  if (!sim) return fail("shf_sim_set_group: null sim");
  if (lanes != 64 && lanes != 32 && lanes != 16) return fail("shf_sim_set_group: lanes must be 16, 32 or 64");
  if (!sim->has_model) return fail("shf_sim_set_group: set the articulation first");
  if (sim->mapping == SHF_MAP_CHAIN) {
    if (lanes != 16 && lanes != 32) return fail("shf_sim_set_group: the chain mapping runs at 16 or 32 lanes per env");
    if (shf_a1_chain_matches(sim->model)) {
      // the chain-mapped fused A1 step takes the knob; gym.simulate / refresh_* stay body-mapped at their own width
      sim->chain_group = lanes;
      return 0;
    }
  }
  if (sim->model.nb + sim->nboxes > lanes || sim->model.nd > lanes)
    return fail("shf_sim_set_group: bodies + box actors (or dofs) exceed the lane group");
  sim->group = lanes;
  return 0;
}
extern "C" int shf_sim_set_mapping(ShfSim* sim, int32_t mapping) {
  if (!sim) return fail("shf_sim_set_mapping: null sim");
  if (mapping != SHF_MAP_BODY && mapping != SHF_MAP_CHAIN && mapping != SHF_MAP_CHAIN_SPLIT)
    return fail("shf_sim_set_mapping: unknown mapping");
  int split = 0;
  if (mapping == SHF_MAP_CHAIN_SPLIT) {
    if (!sim->has_model || !ArmChain<6>::matches(sim->model))
      return fail("shf_sim_set_mapping: the split chain mapping is compiled for a fixed base with one chain of 6 revolute links (the ABB arm)");
    split = 256;   // threads per workgroup: 2 arm waves + 2 box waves for 8 envs (512: 0.100 ms, 128: 0.159 ms, 256: 0.094 ms)
    mapping = SHF_MAP_CHAIN;
  }
  sim->mapping_split = split;
  if (mapping == SHF_MAP_CHAIN) {
    if (!sim->has_model) return fail("shf_sim_set_mapping: set the articulation first");
    const bool a1 = shf_a1_chain_matches(sim->model), arm = ArmChain<6>::matches(sim->model);
    if (!a1 && !arm)
      return fail("shf_sim_set_mapping: the chain mapping needs a floating root with 4 serial chains of 3 revolute links and a "
                  "welded end body (the A1), or a fixed base with one serial chain of 6 revolute links (the ABB arm)");
    if (sim_self(sim) && !a1) return fail("shf_sim_set_mapping: only the A1's chain mapping has a self-collision pass");
    if (a1 && sim->nboxes != 0) return fail("shf_sim_set_mapping: the chain-mapped A1 step supports a single actor per env");
  }
  sim->mapping = mapping;
  return 0;
}

extern "C" int shf_sim_layout(const ShfSim* sim, int32_t id, int64_t shape[4], int32_t* ndim, int32_t* dtype) {
  if (!sim || !sim->finalized) return fail("shf_sim_layout: sim not finalized");
  const int64_t N = sim->n, nd = sim->model.nd, nb = sim->model.nb, A = 1 + sim->nboxes, B = nb + sim->nboxes;
  *dtype = 0;
  shape[0] = shape[1] = shape[2] = shape[3] = 1;
  switch (id) {
    case SHF_T_DOF_STATE: case SHF_T_SIM_DOF: *ndim = 2; shape[0] = N * nd; shape[1] = 2; break;
    case SHF_T_ROOT_STATE: case SHF_T_SIM_ROOT: *ndim = 2; shape[0] = N * A; shape[1] = 13; break;
    case SHF_T_BODY_STATE: *ndim = 2; shape[0] = N * B; shape[1] = 13; break;
    case SHF_T_CONTACT: case SHF_T_SIM_CONTACT: case SHF_T_BODY_FORCE: case SHF_T_BODY_FORCE_POS: *ndim = 2; shape[0] = N * B; shape[1] = 3; break;
    case SHF_T_JACOBIAN:
      *ndim = 4; shape[0] = N; shape[1] = sim->model.fixed_base ? nb - 1 : nb; shape[2] = 6;
      shape[3] = sim->model.fixed_base ? nd : nd + 6; break;
    case SHF_T_EFFORT: case SHF_T_POS_TARGET: case SHF_T_VEL_TARGET: *ndim = 1; shape[0] = N * nd; break;
    case SHF_T_FRICTION: *ndim = 1; shape[0] = N; break;
    case SHF_T_BODY_MASS_SCALE: *ndim = 2; shape[0] = N; shape[1] = nb; break;
    case SHF_T_DROPPED: *ndim = 1; shape[0] = N; *dtype = 1; break;
    case SHF_T_HEIGHTS:
      *dtype = 2;
      if (sim->terr.rows > 0 && sim->terr.warped) {   // samples followed by one byte per vertex (ShfTerrain.warped)
        const int64_t cells = (int64_t)sim->terr.rows * sim->terr.cols;
        *ndim = 1; shape[0] = cells + (cells + 1) / 2;
      } else {
        *ndim = 2; shape[0] = sim->terr.rows > 0 ? sim->terr.rows : 1; shape[1] = sim->terr.rows > 0 ? sim->terr.cols : 1;
      }
      break;
    case SHF_T_MODEL: *ndim = 1; shape[0] = sizeof(ShfModel); *dtype = 3; break;
    case SHF_T_SCENE: *ndim = 1; shape[0] = sizeof(ShfScene); *dtype = 3; break;
    case SHF_T_HULLS: *ndim = 1; shape[0] = sizeof(ShfHullSet); *dtype = 3; break;
    case SHF_T_CONTACT_HIST: *ndim = 2; shape[0] = N; shape[1] = SHF_CONTACT_HIST_BINS + 1; *dtype = 1; break;
    default: return fail("shf_sim_layout: unknown tensor id");
  }
  return 0;
}
extern "C" int shf_sim_bind(ShfSim* sim, int32_t id, void* device_ptr) {
  if (!sim || id < 0 || id >= SHF_T_COUNT) return fail("shf_sim_bind: bad id");
  if (id == SHF_T_MODEL && device_ptr) {
    // the library's own copy -- the caller's model plus the derived fields of shf_model_bounds() -- is what the kernels
    // read: a blob uploaded by a binding that did not derive them is completed here (one blocking copy, at set-up)
    if (!sim->has_model) return fail("shf_sim_bind: SHF_T_MODEL before shf_sim_set_articulation");
    if (hipMemcpy(device_ptr, &sim->model, sizeof(ShfModel), hipMemcpyHostToDevice) != hipSuccess)
      return fail("shf_sim_bind: could not write the model to the bound SHF_T_MODEL buffer");
  }
  if (id == SHF_T_HULLS && device_ptr) {
    if (!sim->has_hulls) return fail("shf_sim_bind: SHF_T_HULLS before shf_sim_set_hulls");
    if (hipMemcpy(device_ptr, &sim->hulls, sizeof(ShfHullSet), hipMemcpyHostToDevice) != hipSuccess)
      return fail("shf_sim_bind: could not write the hulls to the bound SHF_T_HULLS buffer");
  }
  sim->t[id] = device_ptr;
  return 0;
}

static int need(const ShfSim* s, std::initializer_list<int> ids, const char* who) {
  if (!s || !s->finalized) return fail(std::string(who) + ": sim not finalized");
  for (int id : ids) {
    // a model without degrees of freedom has zero-sized dof tensors: the host has no pointer to bind for them
    const bool dof_tensor = id == SHF_T_DOF_STATE || id == SHF_T_SIM_DOF || id == SHF_T_EFFORT || id == SHF_T_POS_TARGET ||
                            id == SHF_T_VEL_TARGET;
    if (!s->t[id] && !(dof_tensor && s->model.nd == 0))
      return fail(std::string(who) + ": tensor " + std::to_string(id) + " not bound");
  }
  return 0;
}

static SimArgs sim_args(const ShfSim* s, bool internal) {
  SimArgs A;
  A.sp = s->sp; A.terr = s->terr;
  A.heights = (const int16_t*)s->t[SHF_T_HEIGHTS];
  A.model = (const ShfModel*)s->t[SHF_T_MODEL];
  A.scene = (const ShfScene*)s->t[SHF_T_SCENE];
  A.nboxes = s->nboxes;
  A.n = s->n;
  A.dof = (float*)s->t[internal ? SHF_T_SIM_DOF : SHF_T_DOF_STATE];
  A.root = (float*)s->t[internal ? SHF_T_SIM_ROOT : SHF_T_ROOT_STATE];
  A.actors = 1 + s->nboxes;
  A.effort = (const float*)s->t[SHF_T_EFFORT];
  A.pos_tgt = (const float*)s->t[SHF_T_POS_TARGET];
  A.vel_tgt = (const float*)s->t[SHF_T_VEL_TARGET];
  A.body_force = nullptr;
  A.body_force_pos = nullptr;
  A.friction = (const float*)s->t[SHF_T_FRICTION];
  A.mscale = (const float*)s->t[SHF_T_BODY_MASS_SCALE];
  A.contact = (float*)s->t[internal ? SHF_T_SIM_CONTACT : SHF_T_CONTACT];
  A.dropped = (int32_t*)s->t[SHF_T_DROPPED];
  A.hulls = (s->model.nhull > 0 && s->model.link_collide != 0) ? (const ShfHullSet*)s->t[SHF_T_HULLS] : nullptr;
  if (s->t[SHF_T_CONTACT_HIST]) {       // the histogram rows carry the drop counters while bound (csrc/shf_task.h: SHF_HIST_FLAG)
    A.dropped = (int32_t*)s->t[SHF_T_CONTACT_HIST];
    A.sp.max_contacts = (A.sp.max_contacts & 0xff) | SHF_HIST_FLAG;
  }
  return A;
}

template <typename K, typename... Args>
static int launch(K kernel, dim3 grid, dim3 block, size_t lds, void* stream, Args... args) {
  if (lds > 160 * 1024) return fail("kernel needs " + std::to_string(lds) + " B of LDS per block, the CU has 160 KiB");
  if (lds > 48 * 1024) {
    // above the default dynamic-LDS limit the kernel has to opt in -- once per (device, kernel, size): the
    // attribute call is not a stream operation and must not recur while the caller's stream is being captured
    // into a hipGraph (shifu_amd/rl captures whole rollouts)
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    const void* fn = reinterpret_cast<const void*>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = granted[{dev, fn}];
    if (have < lds) {
      HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      have = lds;
    }
  }
  hipLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)stream, args...);
  HIP_OK(hipGetLastError());
  return 0;
}

// the same for a kernel of another translation unit, given as its host function pointer and one by-value argument block
static int grant_lds(const void* fn, size_t lds) {
  if (lds > 160 * 1024) return fail("kernel needs " + std::to_string(lds) + " B of LDS per block, the CU has 160 KiB");
  if (lds > 48 * 1024) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = granted[{dev, fn}];
    if (have < lds) {
      HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      have = lds;
    }
  }
  return 0;
}
template <typename ArgBlock>
static int launch_ptr(const void* fn, dim3 grid, dim3 block, size_t lds, void* stream, ArgBlock& A) {
  if (int r = grant_lds(fn, lds)) return r;
  void* args[] = {&A};
  HIP_OK(hipLaunchKernel(fn, grid, block, args, lds, (hipStream_t)stream));
  return 0;
}

// gym.simulate of the shipped arm in the shipped scene under the velocity-level solves has a kernel of its own (k_sim_step_ws_hard:
// arm wave + box wave, the solve regrouped; csrc/shf_kernels.h) -- chosen with the split mapping.  What it is compiled for:
static bool sim_ws_hard_ok(const ShfSim* s) {
  if (!s->finalized || s->sp.solver == SHF_SOLVER_COMPLIANT || s->nboxes == 0 || !sim_plain(s) || sim_self(s) || s->terr.warped) return false;
  if (s->sp.pos_iters < 1 || s->sp.max_contacts > HCK) return false;
  if (!ArmChain<6>::matches(s->model) || !AbbScene::matches(s->nboxes, s->boxes, s->model.nsph)) return false;
  if (!(sim_link(s) ? AbbLinkDims::matches(s->model) : AbbDims::matches(s->model))) return false;
  for (int d = 0; d < s->model.nd; d++)
    if (s->model.drive_mode[d] == SHF_DOF_MODE_VEL) return false;      // (the arm lanes' drive law takes position targets and efforts)
  return true;
}
extern "C" int shf_sim_step_split_supported(const ShfSim* sim) { return (sim && sim_ws_hard_ok(sim)) ? 1 : 0; }

extern "C" int shf_sim_step(ShfSim* sim, void* stream) {
  if (int r = need(sim, {SHF_T_SIM_DOF, SHF_T_SIM_ROOT, SHF_T_SIM_CONTACT, SHF_T_MODEL}, "shf_sim_step")) return r;
  if (sim->model.nhull > 0 && sim->model.link_collide != 0 && sim->nboxes > 0 && !sim->t[SHF_T_HULLS]) return fail("shf_sim_step: the articulation's hulls (SHF_T_HULLS) are not bound");
  if (sim->terr.rows > 0 && !sim->t[SHF_T_HEIGHTS]) return fail("shf_sim_step: heightfield samples not bound");
  SimArgs A = sim_args(sim, true);
  if (sim->force_armed) {
    A.body_force = (const float*)sim->t[SHF_T_BODY_FORCE];
    if (sim->force_at_pos) A.body_force_pos = (const float*)sim->t[SHF_T_BODY_FORCE_POS];
  }
  if (!sim_plain(sim) && sim->nboxes > 0) {
    // a scene with ShfScene.flags or hulls: the run-time-shaped kernels with the convex narrow phase compiled in (csrc/shf_hull.h)
    if (!sim_link(sim) || sim_self(sim)) return fail("shf_sim_step: scene flags / hulls need link contacts (ShfModel.link_collide) and no self-collision");
    if (sim->terr.warped) return fail("shf_sim_step: trimesh terrain with box actors is not supported");
    if (sim->sp.solver != SHF_SOLVER_COMPLIANT) {
      if (sim->sp.max_contacts > HCK || sim->sp.pos_iters < 1) return fail("shf_sim_step: SHF_SOLVER_PGS needs pos_iters >= 1 and max_contacts <= 8");
      if (sim->model.nlevels > HG_LEV || sim->model.nb + sim->nboxes > 32) return fail("shf_sim_step: SHF_SOLVER_PGS: at most 8 tree levels and 32 bodies + box actors");
      sim->force_armed = false; sim->force_at_pos = false;
      const int keep = sim->group;
      sim->group = 32;
      const size_t hlds = sim_lds_bytes(sim, 0, 0, true);
      sim->group = keep;
      return launch(k_sim_step<32, true, false, true, true, true>, dim3((sim->n + 7) / 8), dim3(256), hlds, stream, A);
    }
    sim->force_armed = false; sim->force_at_pos = false;
    if (sim->model.nb + sim->nboxes > sim->group) return fail("shf_sim_step: bodies + boxes exceed the lane group");
    const int epb = 256 / sim->group;
    const dim3 grid((sim->n + epb - 1) / epb), block(256);
    const size_t lds = sim_lds_bytes(sim, 0, 0, true);
    switch (sim->group) {
      case 64: return launch(k_sim_step<64, true, false, true, false, true>, grid, block, lds, stream, A);
      case 32: return launch(k_sim_step<32, true, false, true, false, true>, grid, block, lds, stream, A);
      default: return launch(k_sim_step<16, true, false, true, false, true>, grid, block, lds, stream, A);
    }
  }
  if (sim->sp.solver != SHF_SOLVER_COMPLIANT) {
    // the velocity-level contact solve: built for A1-shaped articulations on their own (csrc/shf_chain_hard.h)
    if (sim->sp.max_contacts > shf_a1_chain_pgs_max_contacts() || sim->sp.pos_iters < 1) return fail("shf_sim_step: SHF_SOLVER_PGS needs pos_iters >= 1 and max_contacts <= 16");
    if (sim->mapping == SHF_MAP_CHAIN && sim->mapping_split && !A.body_force && sim_ws_hard_ok(sim)) {
      const bool link = sim_link(sim);
      const int nbx = sim->nboxes, wepb = 16;
      const int nslots = hard_total_slots(sim->model.np + box_slot_count(nbx, sim_ndyn(sim), sim->model.nsph) + (link ? 2 * SHF_MAX_LINK_CONTACTS : 0), link);
      const size_t wlds = ((size_t)MODEL_WORDS + SCENE_WORDS + ABB_WORDS + STATS_LDS_WORDS +
                           (size_t)wepb * env_lds_words(sim->model.nb + nbx, sim->model.nd, nslots, ABB_TAIL_WORDS(nslots, sim->model.nd) + WS_LINK_STASH_WORDS, 1 + nbx)) * 4;
      const dim3 wgrid((sim->n + wepb - 1) / wepb), wblock(512);
      return link ? launch(k_sim_step_ws_hard<true>, wgrid, wblock, wlds, stream, A) : launch(k_sim_step_ws_hard<false>, wgrid, wblock, wlds, stream, A);
    }
    sim->force_armed = false;
    sim->force_at_pos = false;
    if (sim->nboxes == 0 && shf_a1_chain_matches(sim->model) && !A.body_force_pos)
      return launch_ptr(shf_sim_step_chain_pgs_kernel(sim->terr.warped != 0, sim_self(sim), sim->sp.max_contacts > HCK, sim->sp.solver == SHF_SOLVER_TGS), dim3((sim->n + 7) / 8), dim3(256), shf_sim_step_chain_pgs_lds_bytes(sim_self(sim)), stream, A);
    if (sim->sp.max_contacts > HCK) return fail("shf_sim_step: more than 8 constraints per env (ShfSimParams.max_contacts) are held by the chain-mapped A1 kernels only");
    // any other articulation / a scene with box actors: the body-per-lane sub-step with the generic solve (csrc/shf_hard.h), 32 lanes per env
    if (sim->model.nlevels > HG_LEV) return fail("shf_sim_step: SHF_SOLVER_PGS walks trees of at most 8 levels");
    if (sim->model.nb + sim->nboxes > 32) return fail("shf_sim_step: SHF_SOLVER_PGS runs at 32 lanes per env: at most 32 bodies + box actors");
    if (sim->terr.warped && sim->nboxes > 0) return fail("shf_sim_step: trimesh terrain with box actors is not supported");
    const int keep = sim->group;
    sim->group = 32;
    const size_t hlds = sim_lds_bytes(sim, 0, 0, sim->nboxes > 0);
    sim->group = keep;
    const dim3 hgrid((sim->n + 7) / 8), hblock(256);
    if (sim->nboxes > 0 && sim_link(sim)) {
      const size_t head = ((size_t)MODEL_WORDS + SCENE_WORDS) * 4, env8 = hlds - head, cu = (size_t)160 * 1024;
      if (head + 2 * env8 <= cu && 2 * hlds > cu) {      // sixteen envs per workgroup fit a CU, two workgroups of eight do not
        const dim3 wgrid((sim->n + 15) / 16), wblock(512);
        return sim_self(sim) ? launch(k_sim_step_pgs_wide<true>, wgrid, wblock, head + 2 * env8, stream, A)
                             : launch(k_sim_step_pgs_wide<false>, wgrid, wblock, head + 2 * env8, stream, A);
      }
    }
    if (sim->nboxes > 0) {
      if (sim_link(sim)) return sim_self(sim) ? launch(k_sim_step<32, true, true, true, true>, hgrid, hblock, hlds, stream, A)
                                              : launch(k_sim_step<32, true, false, true, true>, hgrid, hblock, hlds, stream, A);
      return sim_self(sim) ? launch(k_sim_step<32, true, true, false, true>, hgrid, hblock, hlds, stream, A)
                           : launch(k_sim_step<32, true, false, false, true>, hgrid, hblock, hlds, stream, A);
    }
    return sim_self(sim) ? launch(k_sim_step<32, false, true, false, true>, hgrid, hblock, hlds, stream, A)
                         : launch(k_sim_step<32, false, false, false, true>, hgrid, hblock, hlds, stream, A);
  }
  sim->force_armed = false;
  sim->force_at_pos = false;
  const int epb = 256 / sim->group;
  dim3 grid((sim->n + epb - 1) / epb), block(256);
  if (sim->nboxes > 0) {
    if (!sim->t[SHF_T_SCENE]) return fail("shf_sim_step: scene (box actors) not bound");
    if (sim->model.nb + sim->nboxes > sim->group) return fail("shf_sim_step: bodies + boxes exceed the lane group");
    const size_t lds = sim_lds_bytes(sim, 0, 0, true);
    if (sim_link(sim)) {
      // link contacts (ShfModel.link_collide): the run-time-shaped kernels with the candidate pass compiled in
      if (sim_self(sim)) {
        switch (sim->group) {
          case 64: return launch(k_sim_step<64, true, true, true>, grid, block, lds, stream, A);
          case 32: return launch(k_sim_step<32, true, true, true>, grid, block, lds, stream, A);
          default: return launch(k_sim_step<16, true, true, true>, grid, block, lds, stream, A);
        }
      }
      switch (sim->group) {
        case 64: return launch(k_sim_step<64, true, false, true>, grid, block, lds, stream, A);
        case 32: return launch(k_sim_step<32, true, false, true>, grid, block, lds, stream, A);
        default: return launch(k_sim_step<16, true, false, true>, grid, block, lds, stream, A);
      }
    }
    if (sim_self(sim)) {
      switch (sim->group) {
        case 64: return launch(k_sim_step<64, true, true>, grid, block, lds, stream, A);
        case 32: return launch(k_sim_step<32, true, true>, grid, block, lds, stream, A);
        default: return launch(k_sim_step<16, true, true>, grid, block, lds, stream, A);
      }
    }
    switch (sim->group) {
      case 64: return launch(k_sim_step<64, true, false>, grid, block, lds, stream, A);
      case 32: return launch(k_sim_step<32, true, false>, grid, block, lds, stream, A);
      default: return launch(k_sim_step<16, true, false>, grid, block, lds, stream, A);
    }
  }
  const size_t lds = sim_lds_bytes(sim, 0, 0);
  if (sim_self(sim)) {
    switch (sim->group) {
      case 64: return launch(k_sim_step<64, false, true>, grid, block, lds, stream, A);
      case 32: return launch(k_sim_step<32, false, true>, grid, block, lds, stream, A);
      default: return launch(k_sim_step<16, false, true>, grid, block, lds, stream, A);
    }
  }
  switch (sim->group) {
    case 64: return launch(k_sim_step<64, false, false>, grid, block, lds, stream, A);
    case 32: return launch(k_sim_step<32, false, false>, grid, block, lds, stream, A);
    default: return launch(k_sim_step<16, false, false>, grid, block, lds, stream, A);
  }
}

extern "C" int shf_sim_refresh(ShfSim* sim, int32_t mask, void* stream) {
  if (int r = need(sim, {SHF_T_SIM_DOF, SHF_T_SIM_ROOT, SHF_T_MODEL}, "shf_sim_refresh")) return r;
  hipStream_t st = (hipStream_t)stream;
  const size_t N = sim->n, nd = sim->model.nd, nb = sim->model.nb, A = 1 + sim->nboxes, B = nb + sim->nboxes;
  if ((mask & SHF_REFRESH_DOF) && sim->t[SHF_T_DOF_STATE] && nd > 0)
    HIP_OK(hipMemcpyAsync(sim->t[SHF_T_DOF_STATE], sim->t[SHF_T_SIM_DOF], N * nd * 2 * 4, hipMemcpyDeviceToDevice, st));
  if ((mask & SHF_REFRESH_ROOT) && sim->t[SHF_T_ROOT_STATE])
    HIP_OK(hipMemcpyAsync(sim->t[SHF_T_ROOT_STATE], sim->t[SHF_T_SIM_ROOT], N * A * 13 * 4, hipMemcpyDeviceToDevice, st));
  if ((mask & SHF_REFRESH_CONTACT) && sim->t[SHF_T_CONTACT] && sim->t[SHF_T_SIM_CONTACT])
    HIP_OK(hipMemcpyAsync(sim->t[SHF_T_CONTACT], sim->t[SHF_T_SIM_CONTACT], N * B * 3 * 4, hipMemcpyDeviceToDevice, st));
  float* bs = (mask & SHF_REFRESH_BODY) ? (float*)sim->t[SHF_T_BODY_STATE] : nullptr;
  float* jac = ((mask & SHF_REFRESH_JACOBIAN) && sim->model.fixed_base) ? (float*)sim->t[SHF_T_JACOBIAN] : nullptr;
  if (bs || jac) {
    const int epb = 256 / sim->group;
    dim3 grid((sim->n + epb - 1) / epb), block(256);
    const size_t lds = sim_lds_bytes(sim, 0, 0);
    const ShfModel* gm = (const ShfModel*)sim->t[SHF_T_MODEL];
    const float* dof = (const float*)sim->t[SHF_T_SIM_DOF];
    const float* root = (const float*)sim->t[SHF_T_SIM_ROOT];
    switch (sim->group) {
      case 64: return launch(k_body_state<64>, grid, block, lds, stream, gm, sim->n, dof, root, (int)A, bs, jac);
      case 32: return launch(k_body_state<32>, grid, block, lds, stream, gm, sim->n, dof, root, (int)A, bs, jac);
      default: return launch(k_body_state<16>, grid, block, lds, stream, gm, sim->n, dof, root, (int)A, bs, jac);
    }
  }
  return 0;
}

extern "C" int shf_sim_set_dof_command(ShfSim* sim, int32_t tensor_id, const float* values_dev, void* stream) {
  if (tensor_id != SHF_T_EFFORT && tensor_id != SHF_T_POS_TARGET && tensor_id != SHF_T_VEL_TARGET)
    return fail("shf_sim_set_dof_command: tensor_id must be EFFORT, POS_TARGET or VEL_TARGET");
  if (int r = need(sim, {tensor_id}, "shf_sim_set_dof_command")) return r;
  if (!values_dev) return fail("shf_sim_set_dof_command: null values");
  HIP_OK(hipMemcpyAsync(sim->t[tensor_id], values_dev, (size_t)sim->n * sim->model.nd * 4, hipMemcpyDeviceToDevice,
                        (hipStream_t)stream));
  return 0;
}

extern "C" int shf_sim_apply_body_force(ShfSim* sim, const float* force_dev, void* stream) {
  if (int r = need(sim, {SHF_T_BODY_FORCE}, "shf_sim_apply_body_force")) return r;
  if (!force_dev) return fail("shf_sim_apply_body_force: null force tensor");
  HIP_OK(hipMemcpyAsync(sim->t[SHF_T_BODY_FORCE], force_dev, (size_t)sim->n * (sim->model.nb + sim->nboxes) * 3 * 4,
                        hipMemcpyDeviceToDevice, (hipStream_t)stream));
  sim->force_armed = true;
  sim->force_at_pos = false;
  return 0;
}
extern "C" int shf_sim_apply_body_force_at_pos(ShfSim* sim, const float* force_dev, const float* pos_dev, void* stream) {
  if (!pos_dev) return shf_sim_apply_body_force(sim, force_dev, stream);
  if (int r = need(sim, {SHF_T_BODY_FORCE, SHF_T_BODY_FORCE_POS}, "shf_sim_apply_body_force_at_pos")) return r;
  if (!force_dev) return fail("shf_sim_apply_body_force_at_pos: null force tensor");
  const size_t bytes = (size_t)sim->n * (sim->model.nb + sim->nboxes) * 3 * 4;
  HIP_OK(hipMemcpyAsync(sim->t[SHF_T_BODY_FORCE], force_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HIP_OK(hipMemcpyAsync(sim->t[SHF_T_BODY_FORCE_POS], pos_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  sim->force_armed = true;
  sim->force_at_pos = true;
  return 0;
}

extern "C" int shf_sim_commit_root_indexed(ShfSim* sim, const float* root_dev, const int32_t* actor_idx_dev, int32_t n,
                                           void* stream) {
  if (int r = need(sim, {SHF_T_SIM_ROOT}, "shf_sim_commit_root_indexed")) return r;
  if (n <= 0) return 0;
  if (!root_dev || !actor_idx_dev) return fail("shf_sim_commit_root_indexed: null argument");
  return launch(k_commit_rows, dim3(n), dim3(64), 0, stream, root_dev, (float*)sim->t[SHF_T_SIM_ROOT], actor_idx_dev,
                (int)n, 13, 1, 13, (int)(sim->n * (1 + sim->nboxes)));
}
extern "C" int shf_sim_commit_root_all(ShfSim* sim, const float* root_dev, void* stream) {
  if (int r = need(sim, {SHF_T_SIM_ROOT}, "shf_sim_commit_root_all")) return r;
  HIP_OK(hipMemcpyAsync(sim->t[SHF_T_SIM_ROOT], root_dev, (size_t)sim->n * (1 + sim->nboxes) * 13 * 4,
                        hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
extern "C" int shf_sim_commit_dof_indexed(ShfSim* sim, const float* dof_dev, const int32_t* actor_idx_dev, int32_t n,
                                          void* stream) {
  if (int r = need(sim, {SHF_T_SIM_DOF}, "shf_sim_commit_dof_indexed")) return r;
  if (n <= 0) return 0;
  if (!dof_dev || !actor_idx_dev) return fail("shf_sim_commit_dof_indexed: null argument");
  const int nd = sim->model.nd;
  return launch(k_commit_rows, dim3(n), dim3(64), 0, stream, dof_dev, (float*)sim->t[SHF_T_SIM_DOF], actor_idx_dev,
                (int)n, nd * 2, 1 + sim->nboxes, nd * 2, (int)sim->n);
}
extern "C" int shf_sim_commit_reset(ShfSim* sim, const float* root_dev, const int32_t* root_idx_dev, int32_t n_root,
                                    const float* dof_dev, const float* pos_target_dev, const int32_t* dof_actor_idx_dev, int32_t n_dof,
                                    void* stream) {
  if (int r = need(sim, {SHF_T_SIM_ROOT, SHF_T_SIM_DOF, SHF_T_POS_TARGET}, "shf_sim_commit_reset")) return r;
  if (n_root < 0 || n_dof < 0) return fail("shf_sim_commit_reset: negative count");
  if (n_root > 0 && (!root_dev || !root_idx_dev)) return fail("shf_sim_commit_reset: null root argument");
  if (n_dof > 0 && (!dof_actor_idx_dev || (!dof_dev && !pos_target_dev))) return fail("shf_sim_commit_reset: null dof argument");
  if (n_root + 2 * n_dof == 0) return 0;
  return launch(k_commit_reset, dim3(n_root + 2 * n_dof), dim3(32), 0, stream, root_dev, (float*)sim->t[SHF_T_SIM_ROOT], root_idx_dev,
                (int)n_root, (int)(sim->n * (1 + sim->nboxes)), dof_dev, (float*)sim->t[SHF_T_SIM_DOF], pos_target_dev,
                (float*)sim->t[SHF_T_POS_TARGET], dof_actor_idx_dev, (int)n_dof, (int)sim->model.nd, 1 + sim->nboxes, (int)sim->n);
}
extern "C" int shf_sim_set_pos_target_indexed(ShfSim* sim, const float* values_dev, const int32_t* actor_idx_dev,
                                              int32_t n, void* stream) {
  if (int r = need(sim, {SHF_T_POS_TARGET}, "shf_sim_set_pos_target_indexed")) return r;
  if (n <= 0) return 0;
  if (!values_dev || !actor_idx_dev) return fail("shf_sim_set_pos_target_indexed: null argument");
  const int nd = sim->model.nd;
  return launch(k_commit_rows, dim3(n), dim3(64), 0, stream, values_dev, (float*)sim->t[SHF_T_POS_TARGET],
                actor_idx_dev, (int)n, nd, 1 + sim->nboxes, nd, (int)sim->n);
}

extern "C" int shf_sim_reset_all(ShfSim* sim, const float* default_root_dev, const float* default_dof_dev,
                                 const float* env_origins_dev, void* stream) {
  if (int r = need(sim, {SHF_T_SIM_DOF, SHF_T_SIM_ROOT, SHF_T_DOF_STATE, SHF_T_ROOT_STATE, SHF_T_MODEL}, "shf_sim_reset_all"))
    return r;
  if (!default_root_dev || !default_dof_dev) return fail("shf_sim_reset_all: null defaults");
  return launch(k_reset_all, dim3((sim->n + 255) / 256), dim3(256), 0, stream, (const ShfModel*)sim->t[SHF_T_MODEL], sim->n,
                1 + sim->nboxes, default_root_dev, default_dof_dev, env_origins_dev, (float*)sim->t[SHF_T_SIM_DOF],
                (float*)sim->t[SHF_T_DOF_STATE], (float*)sim->t[SHF_T_SIM_ROOT], (float*)sim->t[SHF_T_ROOT_STATE]);
}

// ----------------------------------------------------------- A1 task ABI --
extern "C" int shf_a1_create(ShfSim* sim, const ShfA1TaskParams* params, ShfA1Task** out) {
  if (!sim || !params || !out) return fail("shf_a1_create: null argument");
  if (!sim->finalized) return fail("shf_a1_create: sim not finalized");
  if (sim->nboxes != 0) return fail("shf_a1_create: the fused A1 step supports a single actor per env");
  if (params->num_history != 3) return fail("shf_a1_create: smoothing_action needs num_history == 3");
  if (params->num_height_points > 192) return fail("shf_a1_create: at most 192 height points");
  if (sim->model.nd * params->num_history > 96) return fail("shf_a1_create: action history too large");
  if (sim->model.nb * 13 > SCR_MH) return fail("shf_a1_create: too many bodies for the staging area");
  if (sim->terr.warped && sim->group == 64 && sim->mapping != SHF_MAP_CHAIN)
    return fail("shf_a1_create: a trimesh terrain needs 16 or 32 lanes per env (the 128-VGPR instantiation has no room for it)");
  ShfA1Task* t = new ShfA1Task();
  t->sim = sim;
  t->tp = *params;
  *out = t;
  return 0;
}
extern "C" int shf_a1_destroy(ShfA1Task* task) {
  delete task;
  return 0;
}
extern "C" int shf_a1_layout(const ShfA1Task* task, int32_t id, int64_t shape[4], int32_t* ndim, int32_t* dtype) {
  if (!task) return fail("shf_a1_layout: null task");
  const ShfSim* s = task->sim;
  const int64_t N = s->n, nd = s->model.nd, nb = s->model.nb, H = task->tp.num_history, P = task->tp.num_height_points;
  *dtype = 0;
  shape[0] = shape[1] = shape[2] = shape[3] = 1;
  switch (id) {
    case SHF_A1_ACTIONS: case SHF_A1_TORQUES: *ndim = 2; shape[0] = N; shape[1] = nd; break;
    case SHF_A1_OBS: *ndim = 2; shape[0] = N; shape[1] = 12 + 2 * nd + nd * H + P; break;
    case SHF_A1_REW: *ndim = 1; shape[0] = N; break;
    case SHF_A1_RESET: case SHF_A1_TIMEOUT: *ndim = 1; shape[0] = N; *dtype = 3; break;
    case SHF_A1_EP_LEN: case SHF_A1_LEVELS: case SHF_A1_TYPES: *ndim = 1; shape[0] = N; *dtype = 4; break;
    case SHF_A1_COMMAND: case SHF_A1_ORIGINS: *ndim = 2; shape[0] = N; shape[1] = 3; break;
    case SHF_A1_HISTORY: *ndim = 3; shape[0] = N; shape[1] = nd; shape[2] = H; break;
    case SHF_A1_REW_SUMS: *ndim = 2; shape[0] = 6; shape[1] = N; break;
    case SHF_A1_BASE_VEL: *ndim = 2; shape[0] = N; shape[1] = 9; break;
    case SHF_A1_HEIGHTS: *ndim = 2; shape[0] = N; shape[1] = P; break;
    case SHF_A1_HPOINTS: *ndim = 2; shape[0] = P; shape[1] = 2; break;
    case SHF_A1_PUSH: *ndim = 3; shape[0] = N; shape[1] = nb; shape[2] = 3; break;
    case SHF_A1_TORIGINS: *ndim = 3; shape[0] = task->tp.max_terrain_level; shape[1] = task->tp.num_terrain_cols; shape[2] = 3; break;
    case SHF_A1_RESET_COUNT: *ndim = 1; shape[0] = N; *dtype = 1; break;
    case SHF_A1_DONE_SUMS: *ndim = 2; shape[0] = 8; shape[1] = N; break;
    case SHF_A1_STATS: *ndim = 2; shape[0] = task->stats_ring + 1; shape[1] = 16; break;
    case SHF_A1_STATS_ACC: *ndim = 2; shape[0] = task->stats_ring + 1; shape[1] = STATS_COLS; *dtype = 4; break;
    case SHF_A1_PARAMS: *ndim = 1; shape[0] = sizeof(ShfA1TaskParams); *dtype = 3; break;
    default: return fail("shf_a1_layout: unknown tensor id");
  }
  return 0;
}
extern "C" int shf_a1_bind(ShfA1Task* task, int32_t id, void* device_ptr) {
  if (!task || id < 0 || id >= SHF_A1_COUNT) return fail("shf_a1_bind: bad id");
  task->t[id] = device_ptr;
  return 0;
}

static int a1_args(ShfA1Task* task, const float* raw_actions_dev, const char* who, A1Args& A) {
  if (!task) return fail(std::string(who) + ": null task");
  ShfSim* s = task->sim;
  if (int r = need(s, {SHF_T_DOF_STATE, SHF_T_ROOT_STATE, SHF_T_BODY_STATE, SHF_T_CONTACT, SHF_T_FRICTION, SHF_T_MODEL}, who))
    return r;
  if (s->terr.rows > 0 && !s->t[SHF_T_HEIGHTS]) return fail(std::string(who) + ": heightfield samples not bound");
  for (int id = 0; id < SHF_A1_COUNT; id++)
    if (!task->t[id]) return fail(std::string(who) + ": task tensor " + std::to_string(id) + " not bound");
  A.S = sim_args(s, false);
  A.tp = (const ShfA1TaskParams*)task->t[SHF_A1_PARAMS];
  A.env_off = s->env_off;
  A.raw_actions = raw_actions_dev;
  A.actions = (float*)task->t[SHF_A1_ACTIONS]; A.obs = (float*)task->t[SHF_A1_OBS]; A.rew = (float*)task->t[SHF_A1_REW];
  A.reset = (uint8_t*)task->t[SHF_A1_RESET]; A.timeout = (uint8_t*)task->t[SHF_A1_TIMEOUT];
  A.ep_len = (int64_t*)task->t[SHF_A1_EP_LEN];
  A.command = (float*)task->t[SHF_A1_COMMAND]; A.history = (float*)task->t[SHF_A1_HISTORY];
  A.rew_sums = (float*)task->t[SHF_A1_REW_SUMS]; A.torques = (float*)task->t[SHF_A1_TORQUES];
  A.base_vel = (float*)task->t[SHF_A1_BASE_VEL]; A.heights_out = (float*)task->t[SHF_A1_HEIGHTS];
  A.hpoints = (const float*)task->t[SHF_A1_HPOINTS];
  A.push = (float*)task->t[SHF_A1_PUSH]; A.origins = (float*)task->t[SHF_A1_ORIGINS];
  A.levels = (int64_t*)task->t[SHF_A1_LEVELS]; A.types = (const int64_t*)task->t[SHF_A1_TYPES];
  A.torigins = (const float*)task->t[SHF_A1_TORIGINS];
  A.reset_count = (int32_t*)task->t[SHF_A1_RESET_COUNT];
  A.done_sums = (float*)task->t[SHF_A1_DONE_SUMS];
  A.body_state = (float*)s->t[SHF_T_BODY_STATE];
  A.stats.acc = (unsigned long long*)task->t[SHF_A1_STATS_ACC];
  A.stats.out = (float*)task->t[SHF_A1_STATS];
  A.stats.ring = task->stats_ring;
  A.stats.out_cols = 16;
  return 0;
}

static int a1_step_launch(ShfA1Task* task, const float* raw_actions_dev, void* stream);
extern "C" int shf_a1_step(ShfA1Task* task, const float* raw_actions_dev, void* stream) {
  if (!raw_actions_dev) return fail("shf_a1_step: null actions");
  return a1_step_launch(task, raw_actions_dev, stream);
}
extern "C" int shf_a1_step_random(ShfA1Task* task, void* stream) { return a1_step_launch(task, nullptr, stream); }
static int a1_step_launch(ShfA1Task* task, const float* raw_actions_dev, void* stream) {
  A1Args A;
  if (int r = a1_args(task, raw_actions_dev, "shf_a1_step", A)) return r;
  ShfSim* s = task->sim;
  const int nobs = 12 + 2 * s->model.nd + s->model.nd * task->tp.num_history + task->tp.num_height_points;
  const int epb = 256 / s->group;
  dim3 grid((s->n + epb - 1) / epb), block(256);
  const size_t lds = sim_lds_bytes(s, TASK_WORDS + STATS_LDS_WORDS, SCR_OBS + nobs);
  int r;
  if (s->sp.solver != SHF_SOLVER_COMPLIANT) {
    // the velocity-level contact solve: the chain mapping at two envs per wavefront (csrc/shf_chain_hard.h)
    if (s->mapping != SHF_MAP_CHAIN || s->chain_group != 32 || !shf_a1_chain_matches(s->model))
      return fail("shf_a1_step: ShfSimParams.solver = SHF_SOLVER_PGS runs on the chain mapping at 32 lanes per env (shf_sim_set_mapping)");
    if (s->sp.max_contacts > shf_a1_chain_pgs_max_contacts()) return fail("shf_a1_step: the fused A1 step's solve holds at most 16 constraints per env (ShfSimParams.max_contacts)");
    if (s->sp.pos_iters < 1) return fail("shf_a1_step: ShfSimParams.pos_iters must be >= 1 with SHF_SOLVER_PGS");
    return launch_ptr(shf_a1_chain_pgs_kernel(s->terr.warped != 0, sim_self(s), s->sp.max_contacts > HCK, s->sp.solver == SHF_SOLVER_TGS), dim3((s->n + 7) / 8), block, shf_a1_chain_lds_bytes(32, nobs, sim_self(s)), stream, A);
  }
  if (s->mapping == SHF_MAP_CHAIN) {
    if (!shf_a1_chain_matches(s->model)) return fail("shf_a1_step: the articulation does not fit the chain mapping");
    const void* fn = shf_a1_chain_kernel(s->chain_group, s->terr.warped != 0, sim_self(s));
    if (!fn) return fail(sim_self(s) ? "shf_a1_step: the chain mapping with self-collision runs at 32 lanes per env"
                                     : "shf_a1_step: the chain mapping runs at 16 or 32 lanes per env");
    const int cepb = 256 / s->chain_group;
    return launch_ptr(fn, dim3((s->n + cepb - 1) / cepb), block, shf_a1_chain_lds_bytes(s->chain_group, nobs, sim_self(s)), stream, A);
  }
  if (s->terr.warped && s->group == 64)
    return fail("shf_a1_step: a trimesh terrain needs 16 or 32 lanes per env (the 128-VGPR instantiation has no room for it)");
  if (sim_self(s)) {
    // self-collision: its own instantiations (the capped-VGPR default entry point stays as it is)
    if (s->group == 64) return fail("shf_a1_step: self-collision needs 16 or 32 lanes per env (no register room in the 128-VGPR instantiation)");
    if (A1Dims::matches(s->model)) {
      if (s->group != 32) return fail("shf_a1_step: A1 has 17 bodies, the lane group must be 32 or 64");
      return s->terr.warped ? launch(k_a1_step_self<32, A1Dims>, grid, block, lds, stream, A)
                            : launch(k_a1_step_self_a1_g32, grid, block, lds, stream, A);
    }
    return s->group == 32 ? launch(k_a1_step_self<32, DynDims>, grid, block, lds, stream, A)
                          : launch(k_a1_step_self<16, DynDims>, grid, block, lds, stream, A);
  }
  if (A1Dims::matches(s->model)) {
    switch (s->group) {
      case 64: r = launch(k_a1_step<64, A1Dims>, grid, block, lds, stream, A); break;
      case 32:
        // the capped-VGPR entry point is height-field only; a trimesh terrain takes the generic instantiation
        r = s->terr.warped ? launch(k_a1_step<32, A1Dims>, grid, block, lds, stream, A)
                           : launch(k_a1_step_a1_g32, grid, block, lds, stream, A);
        break;
      default: return fail("shf_a1_step: A1 has 17 bodies, the lane group must be 32 or 64");
    }
  } else {
    switch (s->group) {
      case 64: r = launch(k_a1_step<64, DynDims>, grid, block, lds, stream, A); break;
      case 32: r = launch(k_a1_step<32, DynDims>, grid, block, lds, stream, A); break;
      default: r = launch(k_a1_step<16, DynDims>, grid, block, lds, stream, A); break;
    }
  }
  return r;
}

extern "C" int shf_a1_reset_all(ShfA1Task* task, void* stream) {
  A1Args A;
  if (int r = a1_args(task, nullptr, "shf_a1_reset_all", A)) return r;
  return launch(k_a1_reset_all, dim3((A.S.n + 127) / 128), dim3(128), 0, stream, A);
}

// ---------------------------------------------------------- ABB task ABI --
struct ShfAbbTask {
  ShfSim* sim;
  ShfAbbTaskParams tp;
  void* t[SHF_ABB_COUNT] = {};
  int stats_ring = 256;
};

extern "C" int shf_abb_create(ShfSim* sim, const ShfAbbTaskParams* params, ShfAbbTask** out) {
  if (!sim || !params || !out) return fail("shf_abb_create: null argument");
  if (!sim->finalized) return fail("shf_abb_create: sim not finalized");
  if (!sim->model.fixed_base) return fail("shf_abb_create: the arm must have a fixed base (Jacobian rows ee-1)");
  if (sim->nboxes < 1) return fail("shf_abb_create: the scene needs box actors (table, cube, goal)");
  if (params->cube_actor < 1 || params->cube_actor > sim->nboxes || params->goal_actor < 1 || params->goal_actor > sim->nboxes)
    return fail("shf_abb_create: cube/goal actor index out of range");
  if (params->ee_body < 1 || params->ee_body >= sim->model.nb) return fail("shf_abb_create: bad end-effector body");
  if (sim->model.nd > 6 * 4) return fail("shf_abb_create: too many dofs");
  ShfAbbTask* t = new ShfAbbTask();
  t->sim = sim;
  t->tp = *params;
  *out = t;
  return 0;
}
extern "C" int shf_abb_destroy(ShfAbbTask* task) {
  delete task;
  return 0;
}
extern "C" int shf_abb_layout(const ShfAbbTask* task, int32_t id, int64_t shape[4], int32_t* ndim, int32_t* dtype) {
  if (!task) return fail("shf_abb_layout: null task");
  const int64_t N = task->sim->n, nd = task->sim->model.nd;
  *dtype = 0;
  shape[0] = shape[1] = shape[2] = shape[3] = 1;
  switch (id) {
    case SHF_ABB_ACTIONS: *ndim = 2; shape[0] = N; shape[1] = 3; break;
    case SHF_ABB_OBS: *ndim = 2; shape[0] = N; shape[1] = 6; break;
    case SHF_ABB_REW: *ndim = 1; shape[0] = N; break;
    case SHF_ABB_RESET: case SHF_ABB_TIMEOUT: case SHF_ABB_SUCCESS: *ndim = 1; shape[0] = N; *dtype = 3; break;
    case SHF_ABB_EP_LEN: *ndim = 1; shape[0] = N; *dtype = 4; break;
    case SHF_ABB_REW_SUMS: *ndim = 2; shape[0] = 2; shape[1] = N; break;
    case SHF_ABB_DOF_TARGETS: *ndim = 2; shape[0] = N; shape[1] = nd; break;
    case SHF_ABB_RESET_COUNT: *ndim = 1; shape[0] = N; *dtype = 1; break;
    case SHF_ABB_DONE_SUMS: *ndim = 2; shape[0] = 4; shape[1] = N; break;
    case SHF_ABB_STATS: *ndim = 2; shape[0] = task->stats_ring + 1; shape[1] = 8; break;
    case SHF_ABB_STATS_ACC: *ndim = 2; shape[0] = task->stats_ring + 1; shape[1] = STATS_COLS; *dtype = 4; break;
    case SHF_ABB_PARAMS: *ndim = 1; shape[0] = sizeof(ShfAbbTaskParams); *dtype = 3; break;
    default: return fail("shf_abb_layout: unknown tensor id");
  }
  return 0;
}
extern "C" int shf_abb_bind(ShfAbbTask* task, int32_t id, void* device_ptr) {
  if (!task || id < 0 || id >= SHF_ABB_COUNT) return fail("shf_abb_bind: bad id");
  task->t[id] = device_ptr;
  return 0;
}

static int abb_args(ShfAbbTask* task, const float* raw_actions_dev, const char* who, AbbArgs& A) {
  if (!task) return fail(std::string(who) + ": null task");
  ShfSim* s = task->sim;
  if (int r = need(s, {SHF_T_DOF_STATE, SHF_T_ROOT_STATE, SHF_T_BODY_STATE, SHF_T_CONTACT, SHF_T_JACOBIAN, SHF_T_FRICTION,
                       SHF_T_MODEL, SHF_T_SCENE}, who))
    return r;
  for (int id = 0; id < SHF_ABB_COUNT; id++)
    if (!task->t[id]) return fail(std::string(who) + ": task tensor " + std::to_string(id) + " not bound");
  if (s->model.nhull > 0 && s->model.link_collide != 0 && !s->t[SHF_T_HULLS]) return fail(std::string(who) + ": the articulation's hulls (SHF_T_HULLS) are not bound");
  if (s->model.nb + s->nboxes > s->group) return fail(std::string(who) + ": bodies + boxes exceed the lane group");
  A.S = sim_args(s, false);
  A.tp = (const ShfAbbTaskParams*)task->t[SHF_ABB_PARAMS];
  A.env_off = s->env_off;
  A.raw_actions = raw_actions_dev;
  A.actions = (float*)task->t[SHF_ABB_ACTIONS]; A.obs = (float*)task->t[SHF_ABB_OBS]; A.rew = (float*)task->t[SHF_ABB_REW];
  A.reset = (uint8_t*)task->t[SHF_ABB_RESET]; A.timeout = (uint8_t*)task->t[SHF_ABB_TIMEOUT];
  A.success = (uint8_t*)task->t[SHF_ABB_SUCCESS];
  A.ep_len = (int64_t*)task->t[SHF_ABB_EP_LEN];
  A.rew_sums = (float*)task->t[SHF_ABB_REW_SUMS]; A.dof_targets = (float*)task->t[SHF_ABB_DOF_TARGETS];
  A.reset_count = (int32_t*)task->t[SHF_ABB_RESET_COUNT];
  A.done_sums = (float*)task->t[SHF_ABB_DONE_SUMS];
  A.body_state = (float*)s->t[SHF_T_BODY_STATE];
  A.jacobian = (float*)s->t[SHF_T_JACOBIAN];
  A.stats.acc = (unsigned long long*)task->t[SHF_ABB_STATS_ACC];
  A.stats.out = (float*)task->t[SHF_ABB_STATS];
  A.stats.ring = task->stats_ring;
  A.stats.out_cols = 8;
  return 0;
}

static int abb_step_launch(ShfAbbTask* task, const float* raw_actions_dev, void* stream);
extern "C" int shf_abb_step(ShfAbbTask* task, const float* raw_actions_dev, void* stream) {
  if (!raw_actions_dev) return fail("shf_abb_step: null actions");
  return abb_step_launch(task, raw_actions_dev, stream);
}
extern "C" int shf_abb_step_random(ShfAbbTask* task, void* stream) { return abb_step_launch(task, nullptr, stream); }
// The generic velocity-level step of config 5: sixteen envs per workgroup of 512 threads (k_abb_step_pgs_wide) where their LDS
// fits one CU and eight would leave a CU to a single workgroup (the shipped scene with link contacts: 8.7 KB per env); else
// eight per workgroup of 256 (the rod-only scene: two workgroups per CU).
static bool abb_pgs_wide(const ShfSim* s, size_t* env_bytes, size_t* head_bytes) {
  const int nbx = s->nboxes;
  const int nslots = hard_total_slots(s->model.np + box_slot_count(nbx, sim_ndyn(s), s->model.nsph) + (sim_link(s) ? 2 * SHF_MAX_LINK_CONTACTS : 0), sim_link(s));
  *env_bytes = (size_t)env_lds_words(s->model.nb + nbx, s->model.nd, nslots, ABB_TAIL_WORDS_NOARM(nslots, s->model.nd), 1 + nbx) * 4;
  *head_bytes = ((size_t)MODEL_WORDS + SCENE_WORDS + ABB_WORDS + STATS_LDS_WORDS) * 4;
  const size_t cu = (size_t)160 * 1024;
  return *head_bytes + 16 * *env_bytes <= cu && 2 * (*head_bytes + 8 * *env_bytes) > cu;
}
extern "C" int shf_abb_step_pgs_is_wide(const ShfAbbTask* task) {
  if (!task || !task->sim || task->sim->sp.solver == SHF_SOLVER_COMPLIANT) return 0;
  size_t a, b;
  return abb_pgs_wide(task->sim, &a, &b) ? 1 : 0;
}
static int abb_step_launch(ShfAbbTask* task, const float* raw_actions_dev, void* stream) {
  AbbArgs A;
  if (int r = abb_args(task, raw_actions_dev, "shf_abb_step", A)) return r;
  ShfSim* s = task->sim;
  if (!sim_plain(s)) {
    // a scene with ShfScene.flags or hulls: the run-time-shaped step with the convex narrow phase compiled in (csrc/shf_hull.h)
    if (!sim_link(s)) return fail("shf_abb_step: scene flags / hulls need link contacts (ShfModel.link_collide)");
    if (s->mapping != SHF_MAP_BODY) return fail("shf_abb_step: scene flags / hulls run on the body mapping (shf_sim_set_mapping(SHF_MAP_BODY))");
    if (s->sp.solver != SHF_SOLVER_COMPLIANT) {
      if (s->sp.max_contacts > HCK || s->sp.pos_iters < 1) return fail("shf_abb_step: SHF_SOLVER_PGS needs pos_iters >= 1 and max_contacts <= 8");
      if (s->model.nlevels > HG_LEV || s->model.nb + s->nboxes > 32) return fail("shf_abb_step: SHF_SOLVER_PGS: at most 8 tree levels and 32 bodies + box actors");
      size_t env_bytes, head_bytes;
      abb_pgs_wide(s, &env_bytes, &head_bytes);
      return launch(k_abb_step<32, DynDims, DynScene, true, 0, true, true>, dim3((s->n + 7) / 8), dim3(256), head_bytes + 8 * env_bytes, stream, A);
    }
    const int epbx = 256 / s->group;
    const dim3 gridx((s->n + epbx - 1) / epbx), blockx(256);
    const int nbxx = s->nboxes, nslotsx = s->model.np + box_slot_count(nbxx, sim_ndyn(s), s->model.nsph) + 2 * SHF_MAX_LINK_CONTACTS;
    const size_t ldsx = ((size_t)MODEL_WORDS + SCENE_WORDS + ABB_WORDS + STATS_LDS_WORDS +
                         (size_t)epbx * env_lds_words(s->model.nb + nbxx, s->model.nd, nslotsx, ABB_TAIL_WORDS(nslotsx, s->model.nd), 1 + nbxx)) * 4;
    switch (s->group) {
      case 64: return launch(k_abb_step<64, DynDims, DynScene, true, 0, false, true>, gridx, blockx, ldsx, stream, A);
      case 32: return launch(k_abb_step<32, DynDims, DynScene, true, 0, false, true>, gridx, blockx, ldsx, stream, A);
      default: return launch(k_abb_step<16, DynDims, DynScene, true, 0, false, true>, gridx, blockx, ldsx, stream, A);
    }
  }
  if (s->sp.solver != SHF_SOLVER_COMPLIANT) {
    // the velocity-level contact solve: the run-time-shaped body-per-lane step at 32 lanes per env (csrc/shf_hard.h)
    if (s->sp.max_contacts > HCK || s->sp.pos_iters < 1) return fail("shf_abb_step: SHF_SOLVER_PGS needs pos_iters >= 1 and max_contacts <= 8");
    if (s->model.nlevels > HG_LEV || s->model.nb + s->nboxes > 32) return fail("shf_abb_step: SHF_SOLVER_PGS: at most 8 tree levels and 32 bodies + box actors");
    if (s->mapping == SHF_MAP_CHAIN && s->mapping_split) {
      // the shipped arm in the shipped scene: arm wave + box wave for the free solve and the candidates, the solve regrouped at 32
      // lanes per env (k_abb_step_ws_hard): 512 threads = 16 envs per workgroup
      const bool link = sim_link(s);
      if (!ArmChain<6>::matches(s->model) || !(link ? AbbLinkDims::matches(s->model) : AbbDims::matches(s->model)) ||
          !AbbScene::matches(s->nboxes, s->boxes, s->model.nsph) || s->group != 16)
        return fail("shf_abb_step: the split mapping under SHF_SOLVER_PGS / _TGS needs the shipped arm, the table / cube / pad scene and 16 "
                    "lanes per env");
      const int nbx = s->nboxes, wepb = 16;
      const int nslots = hard_total_slots(s->model.np + box_slot_count(nbx, sim_ndyn(s), s->model.nsph) + (link ? 2 * SHF_MAX_LINK_CONTACTS : 0), link);
      const size_t wlds = ((size_t)MODEL_WORDS + SCENE_WORDS + ABB_WORDS + STATS_LDS_WORDS +
                           (size_t)wepb * env_lds_words(s->model.nb + nbx, s->model.nd, nslots, ABB_TAIL_WORDS(nslots, s->model.nd) + WS_LINK_STASH_WORDS, 1 + nbx)) * 4;
      const dim3 wgrid((s->n + wepb - 1) / wepb), wblock(512);
      return link ? launch(k_abb_step_ws_hard<true>, wgrid, wblock, wlds, stream, A) : launch(k_abb_step_ws_hard<false>, wgrid, wblock, wlds, stream, A);
    }
    size_t env_bytes, head_bytes;
    if (abb_pgs_wide(s, &env_bytes, &head_bytes)) {
      const dim3 wgrid((s->n + 15) / 16), wblock(512);
      return sim_link(s) ? launch(k_abb_step_pgs_wide<true>, wgrid, wblock, head_bytes + 16 * env_bytes, stream, A)
                         : launch(k_abb_step_pgs_wide<false>, wgrid, wblock, head_bytes + 16 * env_bytes, stream, A);
    }
    const size_t lds = head_bytes + 8 * env_bytes;
    const dim3 hgrid((s->n + 7) / 8), hblock(256);
    return sim_link(s) ? launch(k_abb_step<32, DynDims, DynScene, true, 0, true>, hgrid, hblock, lds, stream, A)
                       : launch(k_abb_step<32, DynDims, DynScene, false, 0, true>, hgrid, hblock, lds, stream, A);
  }
  const int epb = 256 / s->group;
  dim3 grid((s->n + epb - 1) / epb), block(256);
  const int nbx = s->nboxes, nslots = s->model.np + box_slot_count(nbx, sim_ndyn(s), s->model.nsph) + (sim_link(s) ? 2 * SHF_MAX_LINK_CONTACTS : 0);
  const size_t lds = ((size_t)MODEL_WORDS + SCENE_WORDS + ABB_WORDS + STATS_LDS_WORDS +
                      (size_t)epb * env_lds_words(s->model.nb + nbx, s->model.nd, nslots, ABB_TAIL_WORDS(nslots, s->model.nd), 1 + nbx)) * 4;
  if (sim_link(s) && s->mapping == SHF_MAP_CHAIN && s->mapping_split) {
    // arm wave + box wave with the link contacts on the box wave: 512 threads = 16 envs per workgroup
    if (!ArmChain<6>::matches(s->model) || !AbbLinkDims::matches(s->model) || !(sim_plain(s) && AbbScene::matches(s->nboxes, s->boxes, s->model.nsph)) || s->group != 16)
      return fail("shf_abb_step: the split mapping with link contacts needs the shipped arm (59 sample points), the table / cube / pad "
                  "scene and 16 lanes per env");
    const int wt = 512, wepb = wt / 32;
    const size_t wlds = ((size_t)MODEL_WORDS + SCENE_WORDS + ABB_WORDS + STATS_LDS_WORDS +
                         (size_t)wepb * env_lds_words(s->model.nb + nbx, s->model.nd, nslots, ABB_TAIL_WORDS(nslots, s->model.nd) + WS_LINK_STASH_WORDS, 1 + nbx)) * 4;
    return launch(k_abb_step_ws<512, true>, dim3((s->n + wepb - 1) / wepb), dim3(wt), wlds, stream, A);
  }
  if (sim_link(s) && s->mapping != SHF_MAP_CHAIN && AbbLinkDims::matches(s->model) && (sim_plain(s) && AbbScene::matches(s->nboxes, s->boxes, s->model.nsph))) {
    // the shipped arm with its link volumes in the shipped scene: compile-time loop bounds, ballot-driven folds
    switch (s->group) {
      case 64: return launch(k_abb_step<64, AbbLinkDims, AbbScene, true>, grid, block, lds, stream, A);
      case 32: return launch(k_abb_step<32, AbbLinkDims, AbbScene, true>, grid, block, lds, stream, A);
      default: return launch(k_abb_step<16, AbbLinkDims, AbbScene, true>, grid, block, lds, stream, A);
    }
  }
  if (sim_link(s) && s->mapping == SHF_MAP_CHAIN)
    return fail("shf_abb_step: the chain mapping (arm recursions on one lane) is compiled without link contacts -- with link contacts use "
                "the split mapping (shf_sim_set_mapping(SHF_MAP_CHAIN_SPLIT), 16 lanes) or the body mapping");
  if (sim_link(s)) {
    switch (s->group) {
      case 64: return launch(k_abb_step<64, DynDims, DynScene, true>, grid, block, lds, stream, A);
      case 32: return launch(k_abb_step<32, DynDims, DynScene, true>, grid, block, lds, stream, A);
      default: return launch(k_abb_step<16, DynDims, DynScene, true>, grid, block, lds, stream, A);
    }
  }
  if (s->mapping == SHF_MAP_CHAIN) {
    // the arm's recursions on one lane (shf_arm.h): compiled for the shipped arm in the shipped scene only
    if (sim_link(s) || !ArmChain<6>::matches(s->model) || !AbbDims::matches(s->model) ||
        !(sim_plain(s) && AbbScene::matches(s->nboxes, s->boxes, s->model.nsph)) || s->group == 64)
      return fail("shf_abb_step: the chain mapping needs the 6-link arm with 3 sample points and one capsule, the table / cube / pad "
                  "scene, no link contacts, and 16 or 32 lanes per env");
    if (s->mapping_split && s->group != 16) return fail("shf_abb_step: the split chain mapping runs at 16 lanes per env");
    if (s->group == 16 && s->mapping_split) {
      // arm and boxes on different waves of the workgroup (k_abb_step_ws): WT / 32 envs per block
      const int wt = s->mapping_split, wepb = wt / 32;
      const size_t wlds = ((size_t)MODEL_WORDS + SCENE_WORDS + ABB_WORDS + STATS_LDS_WORDS +
                           (size_t)wepb * env_lds_words(s->model.nb + nbx, s->model.nd, nslots, ABB_TAIL_WORDS(nslots, s->model.nd), 1 + nbx)) * 4;
      const dim3 wgrid((s->n + wepb - 1) / wepb);
      return launch(k_abb_step_ws<256>, wgrid, dim3(wt), wlds, stream, A);
    }
    return s->group == 32 ? launch(k_abb_step<32, AbbDims, AbbScene, false, 6>, grid, block, lds, stream, A)
                          : launch(k_abb_step<16, AbbDims, AbbScene, false, 6>, grid, block, lds, stream, A);
  }
  if (AbbDims::matches(s->model) && (sim_plain(s) && AbbScene::matches(s->nboxes, s->boxes, s->model.nsph))) {
    switch (s->group) {
      case 64: return launch(k_abb_step<64, AbbDims, AbbScene>, grid, block, lds, stream, A);
      case 32: return launch(k_abb_step<32, AbbDims, AbbScene>, grid, block, lds, stream, A);
      default: return launch(k_abb_step<16, AbbDims, AbbScene>, grid, block, lds, stream, A);
    }
  }
  switch (s->group) {
    case 64: return launch(k_abb_step<64, DynDims, DynScene>, grid, block, lds, stream, A);
    case 32: return launch(k_abb_step<32, DynDims, DynScene>, grid, block, lds, stream, A);
    default: return launch(k_abb_step<16, DynDims, DynScene>, grid, block, lds, stream, A);
  }
}
extern "C" int shf_abb_reset_all(ShfAbbTask* task, void* stream) {
  AbbArgs A;
  if (int r = abb_args(task, nullptr, "shf_abb_reset_all", A)) return r;
  return launch(k_abb_reset_all, dim3((A.S.n + 127) / 128), dim3(128), 0, stream, A);
}

#ifdef SHF_PHASE_CLOCK
// debug builds only (tools/phase_clock.py): read / clear the per-phase cycle counters
extern "C" int shf_debug_phase_cycles(unsigned long long* out, int n, int clear) {
  unsigned long long tmp[48];
  if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(g_phase_cycles), sizeof(tmp)) != hipSuccess) return -1;
  for (int i = 0; i < n && i < 48; i++) out[i] = tmp[i];
  if (clear) {
    std::memset(tmp, 0, sizeof(tmp));
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), tmp, sizeof(tmp)) != hipSuccess) return -1;
  }
  // the chain-mapped kernels live in their own code object with their own counters: add them
  unsigned long long ch[48];
  if (shf_a1_chain_phase_cycles(ch, 48, clear) != 0) return -1;
  for (int i = 0; i < n && i < 48; i++) out[i] += ch[i];
  return 0;
}
#endif
