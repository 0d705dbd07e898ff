// orbfe_ba.hip -- the optimiser entry points: g2o edge evaluation and normal equations (diagnostic), the local BA
// (Optimizer::OptimizeLocalMap, Optimizer.cc:336-391) and Optimizer::OptimizePoseOnly (:33-178).  (Split from orbfe_api.hip in r5.)
#include "orbfe_ctx.h"
extern "C" {

orbfe_status orbfe_ba_eval_edges(orbfe_ctx* c, const orbfe_ba_problem* p, const orbfe_ba_edge_out* o) {
  ApiLock api_lk(c);
  if (!c || !p || !o) return fail(c, ORBFE_EBADARG, "ba_eval_edges: NULL argument");
  const int E = p->n_edges;
  if (E < 0 || p->n_poses < 0 || p->n_points < 0) return fail(c, ORBFE_EBADARG, "ba_eval_edges: negative size");
  if (E == 0) return ORBFE_OK;
  if (!p->poses || !p->points || !p->edge_pose || !p->edge_point || !p->meas || !p->is_stereo || !p->info || !p->huber_delta || !o->error ||
      !o->chi2 || !o->rho)
    return fail(c, ORBFE_EBADARG, "ba_eval_edges: NULL array");
  for (int e = 0; e < E; ++e)
    if (p->edge_pose[e] < 0 || p->edge_pose[e] >= p->n_poses || p->edge_point[e] < 0 || p->edge_point[e] >= p->n_points)
      return fail(c, ORBFE_EBADARG, "ba_eval_edges: edge %d references vertex out of range", e);
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  const size_t o_pose = take((size_t)p->n_poses * 56), o_pt = take((size_t)p->n_points * 24), o_ep = take((size_t)E * 4),
               o_et = take((size_t)E * 4), o_meas = take((size_t)E * 24), o_st = take((size_t)E), o_info = take((size_t)E * 8),
               o_delta = take((size_t)E * 8), o_up_end = take(8), o_err = take((size_t)E * 24), o_chi = take((size_t)E * 8),
               o_rho = take((size_t)E * 16), o_dp = take((size_t)E), o_jpt = take((size_t)E * 72), o_jps = take((size_t)E * 144), o_out_end = take(8);
  TRY(ensure_tmp(c, off));
  uint8_t* b = (uint8_t*)c->d_tmp;
  // up to 16 MB in all: inputs as ONE upload through the page-locked staging buffer and the results as one download (eight copies from
  // and six to pageable memory otherwise -- each staged by the runtime on its own)
  const size_t out_last = o->j_pose ? o_out_end : (o->j_point ? o_jps : o_jpt);
  const bool staged = o_up_end + (out_last - o_err) <= ((size_t)16 << 20);
  uint8_t* hs = nullptr;
  if (staged) {
    TRY(ensure_stage(c, std::max(o_up_end, out_last - o_err)));
    hs = c->main.h_stage;
  }
  auto up = [&](size_t o2, const void* src, size_t bytes) -> hipError_t {
    if (!bytes) return hipSuccess;
    if (staged) {
      std::memcpy(hs + o2, src, bytes);
      return hipSuccess;
    }
    return hipMemcpyAsync(b + o2, src, bytes, hipMemcpyHostToDevice, c->stream);
  };
  HIP_TRY(c, up(o_pose, p->poses, (size_t)p->n_poses * 56));
  HIP_TRY(c, up(o_pt, p->points, (size_t)p->n_points * 24));
  HIP_TRY(c, up(o_ep, p->edge_pose, (size_t)E * 4));
  HIP_TRY(c, up(o_et, p->edge_point, (size_t)E * 4));
  HIP_TRY(c, up(o_meas, p->meas, (size_t)E * 24));
  HIP_TRY(c, up(o_st, p->is_stereo, (size_t)E));
  HIP_TRY(c, up(o_info, p->info, (size_t)E * 8));
  HIP_TRY(c, up(o_delta, p->huber_delta, (size_t)E * 8));
  if (staged) HIP_TRY(c, hipMemcpyAsync(b, hs, o_up_end, hipMemcpyHostToDevice, c->stream));
  BaParamsDev prm = {p->fx, p->fy, p->cx, p->cy, p->bf};
  {
    StageTimer tm(c, ORBFE_STAGE_BA, c->stream);
    launch_ba_edges(c->stream, E, (const double*)(b + o_pose), (const double*)(b + o_pt), (const int32_t*)(b + o_ep),
                    (const int32_t*)(b + o_et), (const double*)(b + o_meas), b + o_st, (const double*)(b + o_info),
                    (const double*)(b + o_delta), prm, (double*)(b + o_err), (double*)(b + o_chi), (double*)(b + o_rho),
                    o->j_point ? (double*)(b + o_jpt) : nullptr, o->j_pose ? (double*)(b + o_jps) : nullptr,
                    o->depth_positive ? b + o_dp : nullptr);
  }
  HIP_TRY(c, hipGetLastError());
  if (staged) {
    HIP_TRY(c, hipMemcpyAsync(hs, b + o_err, out_last - o_err, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drain_timers(c);
    std::memcpy(o->error, hs, (size_t)E * 24);
    std::memcpy(o->chi2, hs + (o_chi - o_err), (size_t)E * 8);
    std::memcpy(o->rho, hs + (o_rho - o_err), (size_t)E * 16);
    if (o->depth_positive) std::memcpy(o->depth_positive, hs + (o_dp - o_err), (size_t)E);
    if (o->j_point) std::memcpy(o->j_point, hs + (o_jpt - o_err), (size_t)E * 72);
    if (o->j_pose) std::memcpy(o->j_pose, hs + (o_jps - o_err), (size_t)E * 144);
    return ORBFE_OK;
  }
  HIP_TRY(c, hipMemcpyAsync(o->error, b + o_err, (size_t)E * 24, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(o->chi2, b + o_chi, (size_t)E * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(o->rho, b + o_rho, (size_t)E * 16, hipMemcpyDeviceToHost, c->stream));
  if (o->j_point) HIP_TRY(c, hipMemcpyAsync(o->j_point, b + o_jpt, (size_t)E * 72, hipMemcpyDeviceToHost, c->stream));
  if (o->j_pose) HIP_TRY(c, hipMemcpyAsync(o->j_pose, b + o_jps, (size_t)E * 144, hipMemcpyDeviceToHost, c->stream));
  if (o->depth_positive) HIP_TRY(c, hipMemcpyAsync(o->depth_positive, b + o_dp, (size_t)E, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  return ORBFE_OK;
}

orbfe_status orbfe_ba_build_system(orbfe_ctx* c, const orbfe_ba_problem* p, const uint8_t* pose_fixed, const orbfe_ba_system_out* o) {
  ApiLock api_lk(c);
  if (!c || !p || !o) return fail(c, ORBFE_EBADARG, "ba_build_system: NULL argument");
  const int E = p->n_edges, NK = p->n_poses, NP = p->n_points;
  if (E < 0 || NK < 0 || NP < 0) return fail(c, ORBFE_EBADARG, "ba_build_system: negative size");
  if (!o->Hpp || !o->bp || !o->Hll || !o->bl) return fail(c, ORBFE_EBADARG, "ba_build_system: NULL output");
  if (E && (!p->poses || !p->points || !p->edge_pose || !p->edge_point || !p->meas || !p->is_stereo || !p->info || !p->huber_delta))
    return fail(c, ORBFE_EBADARG, "ba_build_system: NULL array");
  for (int e = 0; e < E; ++e)
    if (p->edge_pose[e] < 0 || p->edge_pose[e] >= NK || p->edge_point[e] < 0 || p->edge_point[e] >= NP)
      return fail(c, ORBFE_EBADARG, "ba_build_system: edge %d references vertex out of range", e);
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  // vertex -> edges lists, edges in ascending index (counting sort): the summation order of the segmented reductions
  std::vector<int32_t> pt_off(NP + 1, 0), ps_off(NK + 1, 0), pt_edges(std::max(E, 1)), ps_edges(std::max(E, 1));
  for (int e = 0; e < E; ++e) {
    ++pt_off[p->edge_point[e] + 1];
    ++ps_off[p->edge_pose[e] + 1];
  }
  for (int i = 0; i < NP; ++i) pt_off[i + 1] += pt_off[i];
  for (int i = 0; i < NK; ++i) ps_off[i + 1] += ps_off[i];
  {
    std::vector<int32_t> pc(pt_off.begin(), pt_off.end() - 1), kc(ps_off.begin(), ps_off.end() - 1);
    for (int e = 0; e < E; ++e) {
      pt_edges[pc[p->edge_point[e]]++] = e;
      ps_edges[kc[p->edge_pose[e]]++] = e;
    }
  }
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  // r3: the system is built by the kernels of the device-side Levenberg-Marquardt path (k_lm.hip: every edge linearised once, eight lanes
  // per point, a workgroup per pose -- 22 us where round 1's three kernels, each recomputing every edge's Jacobians, took 171); inputs
  // and lists go up as ONE block through the page-locked staging buffer, the blocks come back as one
  const size_t o_pose = take((size_t)NK * 56), o_pt = take((size_t)NP * 24), o_ep = take((size_t)E * 4), o_et = take((size_t)E * 4),
               o_meas = take((size_t)E * 24), o_st = take((size_t)E), o_info = take((size_t)E * 8), o_delta = take((size_t)E * 8),
               o_fix = take((size_t)NK), o_pto = take((size_t)(NP + 1) * 4), o_pte = take((size_t)E * 4), o_pso = take((size_t)(NK + 1) * 4),
               o_pse = take((size_t)E * 4), o_state = take(sizeof(LmState)), o_level = take((size_t)E), o_up_end = take(8),
               o_hpp = take((size_t)NK * 288), o_bp = take((size_t)NK * 48), o_hll = take((size_t)NP * 72),
               o_bl = take((size_t)NP * 24), o_hpl = take((size_t)E * 144), o_out_end = take(8), o_terms = take((size_t)E * 256),
               o_chi = take((size_t)((NP + 31) / 32) * 8);
  TRY(ensure_tmp(c, off));
  TRY(ensure_stage(c, std::max(o_up_end, o_out_end - o_hpp)));
  uint8_t* b = (uint8_t*)c->d_tmp;
  uint8_t* hs = c->main.h_stage;
  auto up = [&](size_t o2, const void* src, size_t bytes) {
    if (bytes) std::memcpy(hs + o2, src, bytes);
  };
  up(o_pose, p->poses, (size_t)NK * 56);
  up(o_pt, p->points, (size_t)NP * 24);
  up(o_ep, p->edge_pose, (size_t)E * 4);
  up(o_et, p->edge_point, (size_t)E * 4);
  up(o_meas, p->meas, (size_t)E * 24);
  up(o_st, p->is_stereo, (size_t)E);
  up(o_info, p->info, (size_t)E * 8);
  up(o_delta, p->huber_delta, (size_t)E * 8);
  if (pose_fixed)
    up(o_fix, pose_fixed, (size_t)NK);
  else
    std::memset(hs + o_fix, 0, (size_t)std::max(NK, 1));
  up(o_pto, pt_off.data(), (size_t)(NP + 1) * 4);
  up(o_pte, pt_edges.data(), (size_t)E * 4);
  up(o_pso, ps_off.data(), (size_t)(NK + 1) * 4);
  up(o_pse, ps_edges.data(), (size_t)E * 4);
  std::memset(hs + o_state, 0, o_up_end - o_state);  // control state (buffer 0 current) and the edge levels (all active)
  HIP_TRY(c, hipMemcpyAsync(b, hs, o_up_end, hipMemcpyHostToDevice, c->stream));
  BaParamsDev prm = {p->fx, p->fy, p->cx, p->cy, p->bf};
  {
    LmLaunch L{};
    L.NK = NK, L.NP = NP, L.E = E, L.nf = 0;
    L.poses[0] = L.poses[1] = (double*)(b + o_pose), L.points[0] = L.points[1] = (double*)(b + o_pt);
    L.terms[0] = L.terms[1] = (double*)(b + o_terms), L.Hpl[0] = L.Hpl[1] = (double*)(b + o_hpl);
    L.Hpp[0] = L.Hpp[1] = (double*)(b + o_hpp), L.bp[0] = L.bp[1] = (double*)(b + o_bp);
    L.Hll[0] = L.Hll[1] = (double*)(b + o_hll), L.bl[0] = L.bl[1] = (double*)(b + o_bl);
    L.chi_part[0] = L.chi_part[1] = (double*)(b + o_chi);
    L.state = (LmState*)(b + o_state);
    L.edge_pose = (const int32_t*)(b + o_ep), L.edge_point = (const int32_t*)(b + o_et);
    L.pt_off = (const int32_t*)(b + o_pto), L.pt_edges = (const int32_t*)(b + o_pte);
    L.ps_off = (const int32_t*)(b + o_pso), L.ps_edges = (const int32_t*)(b + o_pse);
    L.meas = (const double*)(b + o_meas), L.info = (const double*)(b + o_info), L.is_stereo = b + o_st, L.fixed = b + o_fix;
    L.info_eff = (double*)(b + o_info), L.delta_eff = (double*)(b + o_delta), L.chi2_last = nullptr, L.level = b + o_level;
    L.prm = prm;
    StageTimer tm(c, ORBFE_STAGE_BA, c->stream);
    launch_lm_build(c->stream, L, 0, 0, 0, true);
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(hs, b + o_hpp, (o->Hpl ? o_out_end : o_hpl) - o_hpp, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  std::memcpy(o->Hpp, hs, (size_t)NK * 288);
  std::memcpy(o->bp, hs + (o_bp - o_hpp), (size_t)NK * 48);
  std::memcpy(o->Hll, hs + (o_hll - o_hpp), (size_t)NP * 72);
  std::memcpy(o->bl, hs + (o_bl - o_hpp), (size_t)NP * 24);
  if (o->Hpl) std::memcpy(o->Hpl, hs + (o_hpl - o_hpp), (size_t)E * 144);
  return ORBFE_OK;
}

// Optimizer::OptimizeLocalMap's two optimize() calls (Optimizer.cc:336-362) with g2o's Levenberg-Marquardt control on the host
// (a handful of scalars per trial) and every vertex / edge / block operation on the device.
orbfe_status orbfe_ba_local_optimize(orbfe_ctx* c, const orbfe_ba_problem* p, const uint8_t* pose_fixed, int32_t iters_first,
                                     int32_t iters_second, const volatile uint8_t* stop_flag, const orbfe_ba_optimize_out* o) {
  ApiLock api_lk(c);
  static const bool trace_host = getenv("ORBFE_LBA_TRACE") != nullptr;  // diagnostic: host phases of this call on stderr
  auto t_prev = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!trace_host) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[orbfe lba] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_prev).count());
    t_prev = now;
  };
  if (!c || !p || !o) return fail(c, ORBFE_EBADARG, "ba_local_optimize: NULL argument");
  const int E = p->n_edges, NK = p->n_poses, NP = p->n_points;
  if (E < 0 || NK < 0 || NP < 0 || iters_first < 0 || iters_second < 0) return fail(c, ORBFE_EBADARG, "ba_local_optimize: negative size");
  if (!o->poses || !o->points) return fail(c, ORBFE_EBADARG, "ba_local_optimize: NULL output");
  if ((NK && !p->poses) || (NP && !p->points) ||
      (E && (!p->edge_pose || !p->edge_point || !p->meas || !p->is_stereo || !p->info || !p->huber_delta)))
    return fail(c, ORBFE_EBADARG, "ba_local_optimize: NULL array");
  for (int e = 0; e < E; ++e)
    if (p->edge_pose[e] < 0 || p->edge_pose[e] >= NK || p->edge_point[e] < 0 || p->edge_point[e] >= NP)
      return fail(c, ORBFE_EBADARG, "ba_local_optimize: edge %d references vertex out of range", e);
  // free poses, vertex -> edges lists (ascending edge index), pose-pair lists of the Schur complement
  std::vector<int32_t> slot(std::max(NK, 1), -1), free_pose;
  for (int k = 0; k < NK; ++k)
    if (!(pose_fixed && pose_fixed[k])) {
      slot[k] = (int32_t)free_pose.size();
      free_pose.push_back(k);
    }
  const int nf = (int)free_pose.size();
  // (no bound on nf: up to LBA_MAX_FREE free keyframes the reduced system is factorised by one workgroup out of LDS, beyond that by the
  //  multi-workgroup path of k_lba.hip with its panel in global memory)
  std::vector<int32_t> pt_off(NP + 1, 0), ps_off(NK + 1, 0), pt_edges(std::max(E, 1)), ps_edges(std::max(E, 1));
  for (int e = 0; e < E; ++e) {
    ++pt_off[p->edge_point[e] + 1];
    ++ps_off[p->edge_pose[e] + 1];
  }
  for (int i = 0; i < NP; ++i) pt_off[i + 1] += pt_off[i];
  for (int i = 0; i < NK; ++i) ps_off[i + 1] += ps_off[i];
  {
    std::vector<int32_t> pc(pt_off.begin(), pt_off.end() - 1), kc(ps_off.begin(), ps_off.end() - 1);
    for (int e = 0; e < E; ++e) {
      pt_edges[pc[p->edge_point[e]]++] = e;
      ps_edges[kc[p->edge_pose[e]]++] = e;
    }
  }
  // The device-side Levenberg-Marquardt path (k_lm.hip) builds the pair lists of the reduced system itself, from a (pose, point) -> edge
  // table: that needs a pose to observe a point at most once (as every map of the reference does); anything else takes the host-driven path.
  bool single_obs = true;
  int pair_cap = 1;
  {
    std::vector<int32_t> seen(std::max(NK, 1), -1);
    for (int pt = 0; pt < NP && single_obs; ++pt)
      for (int a = pt_off[pt]; a < pt_off[pt + 1]; ++a) {
        const int k = p->edge_pose[pt_edges[a]];
        if (seen[k] == pt) {
          single_obs = false;
          break;
        }
        seen[k] = pt;
      }
    for (int k = 0; k < NK; ++k)
      if (slot[k] >= 0) pair_cap = std::max(pair_cap, ps_off[k + 1] - ps_off[k]);
  }
  mark("validate + vertex lists");
  std::vector<int32_t> pair_off(1, 0);
  std::vector<int2> pairs;
  const bool lower_only = c && c->lm_on_device && E > 0 && nf <= LM_BIG_MAX_NB && single_obs;  // (= dev_lm below)
  const bool big_solver = lower_only && nf > LM_CHOL_MAX_NB;  // the blocked multi-workgroup Cholesky of k_lmbig.hip
  if (!lower_only) {
    pair_off.assign((size_t)nf * nf + 1, 0);
    for (int pt = 0; pt < NP; ++pt)
      for (int a = pt_off[pt]; a < pt_off[pt + 1]; ++a) {
        const int i = slot[p->edge_pose[pt_edges[a]]];
        if (i < 0) continue;
        for (int b2 = pt_off[pt]; b2 < pt_off[pt + 1]; ++b2) {
          const int j = slot[p->edge_pose[pt_edges[b2]]];
          if (j >= 0) ++pair_off[(size_t)i * nf + j + 1];
        }
      }
    for (size_t q = 0; q < (size_t)nf * nf; ++q) pair_off[q + 1] += pair_off[q];
    pairs.resize(std::max<size_t>(pair_off.back(), 1));
    std::vector<int32_t> cur(pair_off.begin(), pair_off.end() - 1);
    for (int pt = 0; pt < NP; ++pt)
      for (int a = pt_off[pt]; a < pt_off[pt + 1]; ++a) {
        const int e1 = pt_edges[a], i = slot[p->edge_pose[e1]];
        if (i < 0) continue;
        for (int b2 = pt_off[pt]; b2 < pt_off[pt + 1]; ++b2) {
          const int e2 = pt_edges[b2], j = slot[p->edge_pose[e2]];
          if (j >= 0) pairs[cur[(size_t)i * nf + j]++] = make_int2(e1, e2);
        }
      }
  }
  mark("pair lists");
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  const size_t n = (size_t)6 * nf;
  const size_t o_pose = take((size_t)NK * 56), o_pt = take((size_t)NP * 24), o_pose_bk = take((size_t)NK * 56), o_pt_bk = take((size_t)NP * 24),
               o_ep = take((size_t)E * 4), o_et = take((size_t)E * 4), o_meas = take((size_t)E * 24), o_st = take((size_t)E),
               o_info = take((size_t)E * 8), o_info_eff = take((size_t)E * 8), o_delta = take((size_t)E * 8), o_fix = take((size_t)NK),
               o_pto = take((size_t)(NP + 1) * 4), o_pte = take((size_t)E * 4), o_pso = take((size_t)(NK + 1) * 4), o_pse = take((size_t)E * 4),
               o_free = take((size_t)nf * 4), o_slot = take((size_t)NK * 4), o_pairoff = take(pair_off.size() * 4),
               o_pairs = take(pairs.size() * 8), o_lmstate = take(sizeof(LmState)),  // (the initial control state rides in the one upload)
               o_hpp = take((size_t)NK * 288), o_bp = take((size_t)NK * 48), o_hll = take((size_t)NP * 72),
               o_bl = take((size_t)NP * 24), o_hpl = take((size_t)E * 144), o_w = take((size_t)E * 144),
               o_s = take(n * n * 8), o_rhs = take(n * 8), o_x = take((n + 48) * 8), o_dxp = take((size_t)NK * 48), o_dxl = take((size_t)NP * 24),
               o_err = take((size_t)E * 24), o_chi2 = take((size_t)E * 8), o_rho = take((size_t)E * 16),
               // one block that starts as zeros (ONE fill): edge levels | chi2 of the last linearisation | point inverses
               o_level = take((size_t)E), o_last = take((size_t)E * 8), o_dinv = take((size_t)NP * 72), o_zero_end = take(8),
               o_depth = take((size_t)E), o_bad = take((size_t)E), o_sc = take(64),
               o_big = take(nf > LBA_MAX_FREE ? ((n + 1) * 6 + (size_t)nf * 36 + n) * 8 : 8);
  // the device-side Levenberg-Marquardt path (k_lm.hip): second estimate / system buffers, per-edge terms, blocked reduced system
  const bool dev_lm = lower_only;
  const int chi_blocks = (NP + 31) / 32, scale_blocks = (NP + 31) / 32 + (NK + 255) / 256;  // (k_lm_linpoints: a partial sum per block of 32 points)
  size_t l_ptable = 0, l_pairs = 0, l_paircnt = 0, l_pose1 = 0, l_pt1 = 0, l_terms[2] = {0, 0}, l_hpl1 = 0, l_hpp1 = 0, l_bp1 = 0, l_hll1 = 0, l_bl1 = 0, l_chi[2] = {0, 0}, l_sblk = 0,
         l_scale = 0, l_big = 0, l_bigflags = 0, l_biginv = 0, l_pose_out = 0, l_pt_out = 0, l_chi2_out = 0, l_level_out = 0, l_bad_out = 0, l_state_out = 0, l_out_end = 0;
  if (dev_lm) {
    l_pose1 = take((size_t)NK * 56), l_pt1 = take((size_t)NP * 24);
    l_ptable = take((size_t)nf * NP * 4), l_pairs = take((size_t)nf * (nf + 1) / 2 * pair_cap * 8), l_paircnt = take((size_t)nf * (nf + 1) / 2 * 4);
    l_terms[0] = take((size_t)E * 256), l_terms[1] = take((size_t)E * 256);
    l_hpl1 = take((size_t)E * 144), l_hpp1 = take((size_t)NK * 288), l_bp1 = take((size_t)NK * 48), l_hll1 = take((size_t)NP * 72),
    l_bl1 = take((size_t)NP * 24);
    l_chi[0] = take((size_t)chi_blocks * 8), l_chi[1] = take((size_t)chi_blocks * 8);
    l_sblk = take(big_solver ? 8 : (size_t)nf * (nf + 1) / 2 * 288), l_scale = take((size_t)scale_blocks * 8);
    l_big = take(big_solver ? lm_big_bytes(nf) : 8), l_bigflags = take(big_solver ? (2 * ((size_t)lm_big_ld(nf) / 48) + 4) * 4 : 8),
    l_biginv = take(big_solver ? lm_big_inv_bytes(nf) : 8);
    // the results as ONE block (one download): poses | points | chi2 | level | bad
    l_pose_out = take((size_t)NK * 56), l_pt_out = take((size_t)NP * 24), l_chi2_out = take((size_t)E * 8), l_level_out = take((size_t)E),
    l_bad_out = take((size_t)E), l_state_out = take(sizeof(LmState)), l_out_end = take(8);  // (+ the control state as the last control step left it)
  }
  TRY(ensure_tmp(c, off));
  uint8_t* b = (uint8_t*)c->d_tmp;
  hipStream_t st = c->stream;
  // ONE upload: the inputs and the lists built above are laid out in page-locked staging memory exactly as in the device scratch
  // (they are its first o_hpp bytes) and go up as a single asynchronous copy -- eighteen copies from pageable memory were staged by the
  // runtime one by one, ~0.3 ms of a 4 ms call
  const size_t up_bytes = o_hpp;
  TRY(ensure_stage(c, std::max(up_bytes, dev_lm ? l_out_end - l_pose_out : (size_t)0)));  // (also the target of the one result download)
  uint8_t* hs = c->main.h_stage;
  auto up = [&](size_t o2, const void* src, size_t bytes) -> hipError_t {
    if (bytes) std::memcpy(hs + o2, src, bytes);
    return hipSuccess;
  };
  std::vector<uint8_t> fixed_h(std::max(NK, 1), 0);
  if (pose_fixed) std::memcpy(fixed_h.data(), pose_fixed, NK);
  HIP_TRY(c, up(o_pose, p->poses, (size_t)NK * 56));
  HIP_TRY(c, up(o_pt, p->points, (size_t)NP * 24));
  HIP_TRY(c, up(o_ep, p->edge_pose, (size_t)E * 4));
  HIP_TRY(c, up(o_et, p->edge_point, (size_t)E * 4));
  HIP_TRY(c, up(o_meas, p->meas, (size_t)E * 24));
  HIP_TRY(c, up(o_st, p->is_stereo, (size_t)E));
  HIP_TRY(c, up(o_info, p->info, (size_t)E * 8));
  HIP_TRY(c, up(o_info_eff, p->info, (size_t)E * 8));
  HIP_TRY(c, up(o_delta, p->huber_delta, (size_t)E * 8));
  HIP_TRY(c, up(o_fix, fixed_h.data(), (size_t)NK));
  HIP_TRY(c, up(o_pto, pt_off.data(), (size_t)(NP + 1) * 4));
  HIP_TRY(c, up(o_pte, pt_edges.data(), (size_t)E * 4));
  HIP_TRY(c, up(o_pso, ps_off.data(), (size_t)(NK + 1) * 4));
  HIP_TRY(c, up(o_pse, ps_edges.data(), (size_t)E * 4));
  HIP_TRY(c, up(o_free, free_pose.data(), (size_t)nf * 4));
  HIP_TRY(c, up(o_slot, slot.data(), (size_t)NK * 4));
  HIP_TRY(c, up(o_pairoff, pair_off.data(), pair_off.size() * 4));
  HIP_TRY(c, up(o_pairs, pairs.data(), pairs.size() * 8));
  {
    LmState init{};
    init.iters[0] = iters_first, init.iters[1] = iters_second, init.need_chi = 1, init.ok = 1;
    HIP_TRY(c, up(o_lmstate, &init, sizeof init));
  }
  mark("stage inputs");
  HIP_TRY(c, hipMemcpyAsync(b, hs, up_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemsetAsync(b + o_level, 0, o_zero_end - o_level, st));  // (every memset is a launch of 4.6 us: six of them preceded the first kernel)

  const BaParamsDev prm = {p->fx, p->fy, p->cx, p->cy, p->bf};
  double* d_poses = (double*)(b + o_pose);
  double* d_points = (double*)(b + o_pt);
  const int32_t* d_ek = (const int32_t*)(b + o_ep);
  const int32_t* d_ep = (const int32_t*)(b + o_et);
  double* d_sc = (double*)(b + o_sc);  // [0] robust chi2, [1] max diagonal, [2] lambda, [3] ok (int), [4] scale
  if (dev_lm) {
    // ---- Levenberg-Marquardt control on the device: enqueue the whole optimisation, synchronise once ------------------------------
    if (!c->h_abort) {
      HIP_TRY(c, hipHostMalloc((void**)&c->h_abort, 64, hipHostMallocMapped));
    }
    void* d_abort = nullptr;
    HIP_TRY(c, hipHostGetDevicePointer(&d_abort, (void*)c->h_abort, 0));
    *c->h_abort = (stop_flag && *stop_flag) ? 1 : 0;
    LmLaunch L{};
    L.NK = NK, L.NP = NP, L.E = E, L.nf = nf;
    L.poses[0] = d_poses, L.poses[1] = (double*)(b + l_pose1), L.points[0] = d_points, L.points[1] = (double*)(b + l_pt1);
    L.terms[0] = (double*)(b + l_terms[0]), L.terms[1] = (double*)(b + l_terms[1]);
    L.Hpl[0] = (double*)(b + o_hpl), L.Hpl[1] = (double*)(b + l_hpl1), L.Hpp[0] = (double*)(b + o_hpp), L.Hpp[1] = (double*)(b + l_hpp1);
    L.bp[0] = (double*)(b + o_bp), L.bp[1] = (double*)(b + l_bp1), L.Hll[0] = (double*)(b + o_hll), L.Hll[1] = (double*)(b + l_hll1);
    L.bl[0] = (double*)(b + o_bl), L.bl[1] = (double*)(b + l_bl1), L.chi_part[0] = (double*)(b + l_chi[0]), L.chi_part[1] = (double*)(b + l_chi[1]);
    L.state = (LmState*)(b + o_lmstate);
    L.edge_pose = d_ek, L.edge_point = d_ep, L.pt_off = (const int32_t*)(b + o_pto), L.pt_edges = (const int32_t*)(b + o_pte);
    L.ps_off = (const int32_t*)(b + o_pso), L.ps_edges = (const int32_t*)(b + o_pse), L.free_pose = (const int32_t*)(b + o_free);
    L.pose_slot = (const int32_t*)(b + o_slot), L.pairs = (int2*)(b + l_pairs), L.pair_cnt = (int32_t*)(b + l_paircnt);
    L.pair_table = (int32_t*)(b + l_ptable), L.pair_cap = pair_cap;
    L.meas = (const double*)(b + o_meas), L.info = (const double*)(b + o_info), L.is_stereo = b + o_st, L.fixed = b + o_fix;
    L.info_eff = (double*)(b + o_info_eff), L.delta_eff = (double*)(b + o_delta), L.chi2_last = (double*)(b + o_last), L.level = b + o_level;
    L.Dinv = (double*)(b + o_dinv), L.W = (double*)(b + o_w), L.Sblk = (double*)(b + l_sblk), L.rhs = (double*)(b + o_rhs), L.x = (double*)(b + o_x);
    L.scale_part = (double*)(b + l_scale), L.chi2_out = (double*)(b + l_chi2_out), L.poses_out = (double*)(b + l_pose_out);
    L.points_out = (double*)(b + l_pt_out), L.bad = b + l_bad_out, L.level_out = b + l_level_out;
    L.abort_flag = (const volatile uint8_t*)d_abort, L.prm = prm;
    // (the initial state went up with the inputs; the ticket and the point inverses -- read by a trial whose point block was singular --
    //  are part of the one zero fill)
    L.state_out = (LmState*)(b + l_state_out);
    L.M = big_solver ? (double*)(b + l_big) : nullptr, L.ld = big_solver ? lm_big_ld(nf) : 0, L.lmb_flags = (int32_t*)(b + l_bigflags), L.lmb_inv = (double*)(b + l_biginv);
    StageTimer tm(c, ORBFE_STAGE_BA, st);
    if (big_solver) {
      HIP_TRY(c, hipMemsetAsync(b + l_big, 0, (l_bigflags - l_big) + (2 * ((size_t)L.ld / 48) + 4) * 4, st));  // the matrix and the flags behind it
      launch_lm_big_init(st, L);
    }
    HIP_TRY(c, hipMemsetAsync(L.pair_table, 0xFF, (size_t)nf * NP * 4, st));
    launch_lm_pairs(st, L);
    launch_lm_build(st, L, 0, 0, iters_first > 0 ? 1 : 0, true);  // computeActiveErrors + buildSystem at the initial estimate
    launch_lm_maxdiag(st, L, 0);
    // trials provisioned per pass: every iteration needs at least one, a rejected trial costs one more; what is left over runs as no-ops
    // (a few microseconds each), what is missing is enqueued in the next pass, after the one synchronisation of this one
    // (measured: a provisioned trial that turns out not to be needed is six empty launches of 4.6 us; one spare
    // -- in round 0, where coming up one short would leave the ten trials of round 1 as no-ops in this pass; round 1 gets none: if a trial
    // of it is rejected, the second pass enqueues what is missing)
    int steps_a = std::min(iters_first + 1, 24), steps_b = std::min(iters_second, 24);
    LmState fin{};
    for (int pass = 0;; ++pass) {
      launch_lm_steps(st, L, steps_a);
      launch_lm_switch(st, L);
      launch_lm_steps(st, L, steps_b);
      launch_lm_final(st, L);
      HIP_TRY(c, hipGetLastError());
      const size_t out_bytes = l_out_end - l_pose_out;
      HIP_TRY(c, hipMemcpyAsync(hs, b + l_pose_out, out_bytes, hipMemcpyDeviceToHost, st));  // the upload from hs finished long ago (stream order)
      if (stop_flag) {
        // the device polls the mapped byte between the trials; the caller's flag (LocalMapping::mbAbortBA, written by the Tracking
        // thread) is mirrored into it while this thread waits
        hipEvent_t ev = nullptr;
        HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t er = hipEventRecord(ev, st);
        while (er == hipSuccess) {
          if (*stop_flag) *c->h_abort = 1;
          er = hipEventQuery(ev);
          if (er == hipErrorNotReady) {
            (void)hipGetLastError();
            er = hipSuccess;
            sched_yield();
            continue;
          }
          break;
        }
        (void)hipEventDestroy(ev);
        HIP_TRY(c, er);
      }
      mark("enqueue");
      HIP_TRY(c, hipStreamSynchronize(st));
      mark("device (wait)");
      std::memcpy(&fin, hs + (l_state_out - l_pose_out), sizeof fin);
      if (fin.finalized) {
        std::memcpy(o->poses, hs, (size_t)NK * 56);
        std::memcpy(o->points, hs + (l_pt_out - l_pose_out), (size_t)NP * 24);
        if (o->chi2) std::memcpy(o->chi2, hs + (l_chi2_out - l_pose_out), (size_t)E * 8);
        if (o->level) std::memcpy(o->level, hs + (l_level_out - l_pose_out), (size_t)E);
        if (o->bad) std::memcpy(o->bad, hs + (l_bad_out - l_pose_out), (size_t)E);
        break;
      }
      if (pass >= 4096) return fail(c, ORBFE_EDEVICE, "ba_local_optimize: the device-side Levenberg-Marquardt loop did not finish (round %d, phase %d)", fin.round, fin.phase);
      // more trials were needed than provisioned: continue where the state stands
      steps_a = fin.switched || fin.round == 2 ? 0 : std::min(std::max(iters_first - fin.it, 0) + 2, 24);
      steps_b = std::min((fin.switched ? std::max(iters_second - fin.it, 0) : iters_second) + 2, 24);
    }
    if (o->iterations) {
      o->iterations[0] = fin.done[0];
      o->iterations[1] = fin.done[1];
    }
    drain_timers(c);
    mark("results out");
    return ORBFE_OK;
  }
  struct HostScalars {
    double chi, maxdiag, lambda;
    int32_t ok, pad;
    double scale;
  };
  auto evaluate = [&](const double* d_info) {  // computeActiveErrors + activeRobustChi2
    launch_ba_edges(st, E, d_poses, d_points, d_ek, d_ep, (const double*)(b + o_meas), b + o_st, d_info, (const double*)(b + o_delta), prm,
                    (double*)(b + o_err), (double*)(b + o_chi2), (double*)(b + o_rho), nullptr, nullptr, b + o_depth);
    launch_lba_chi2_sum(st, E, (const double*)(b + o_chi2), (const double*)(b + o_rho), b + o_level, (double*)(b + o_last), d_sc);
  };
  auto read_scalars = [&](HostScalars& h) -> hipError_t {
    hipError_t e = hipMemcpyAsync(&h, d_sc, sizeof h, hipMemcpyDeviceToHost, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
  };
  auto stopped = [&]() { return stop_flag && *stop_flag; };
  StageTimer tm(c, ORBFE_STAGE_BA, st);
  auto optimize = [&](int iterations, int32_t& done) -> orbfe_status {  // SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve
    done = 0;
    if (E == 0) return ORBFE_OK;
    double lambda = 0, ni = 2;
    for (int it = 0; it < iterations; ++it) {
      if (stopped()) break;
      ++done;
      evaluate((const double*)(b + o_info_eff));
      launch_ba_system(st, NK, NP, E, d_poses, d_points, d_ek, d_ep, (const double*)(b + o_meas), b + o_st, (const double*)(b + o_info_eff),
                       (const double*)(b + o_delta), prm, b + o_fix, (const int32_t*)(b + o_pto), (const int32_t*)(b + o_pte),
                       (const int32_t*)(b + o_pso), (const int32_t*)(b + o_pse), (double*)(b + o_hpp), (double*)(b + o_bp),
                       (double*)(b + o_hll), (double*)(b + o_bl), (double*)(b + o_hpl));
      if (it == 0) launch_lba_maxdiag(st, NK, NP, (const double*)(b + o_hpp), (const double*)(b + o_hll), b + o_fix, d_sc + 1);
      HostScalars h;
      HIP_TRY(c, read_scalars(h));
      double current_chi = h.chi;
      if (it == 0) {
        lambda = 1e-5 * h.maxdiag;  // computeLambdaInit, tau = 1e-5
        ni = 2;
      }
      double rho = 0;
      int qmax = 0;
      do {
        HIP_TRY(c, hipMemcpyAsync(b + o_pose_bk, d_poses, (size_t)NK * 56, hipMemcpyDeviceToDevice, st));  // push()
        HIP_TRY(c, hipMemcpyAsync(b + o_pt_bk, d_points, (size_t)NP * 24, hipMemcpyDeviceToDevice, st));
        struct {
          double lambda;
          int32_t ok, pad;
        } upv = {lambda, 1, 0};
        HIP_TRY(c, hipMemcpyAsync(d_sc + 2, &upv, sizeof upv, hipMemcpyHostToDevice, st));
        launch_lba_solve(st, NK, NP, E, nf, (const int32_t*)(b + o_free), (const int32_t*)(b + o_slot), (const int32_t*)(b + o_pairoff),
                         (const int2*)(b + o_pairs), (const int32_t*)(b + o_pso), (const int32_t*)(b + o_pse), (const int32_t*)(b + o_pto),
                         (const int32_t*)(b + o_pte), d_ek, d_ep, b + o_fix, (const double*)(b + o_hpp), (const double*)(b + o_bp),
                         (const double*)(b + o_hll), (const double*)(b + o_bl), (const double*)(b + o_hpl), d_sc + 2, (double*)(b + o_dinv),
                         (double*)(b + o_w), (double*)(b + o_s), (double*)(b + o_rhs), (double*)(b + o_x), (int*)(d_sc + 3), d_poses, d_points,
                         (double*)(b + o_dxp), (double*)(b + o_dxl), d_sc + 4, (double*)(b + o_big));
        evaluate((const double*)(b + o_info_eff));
        HIP_TRY(c, read_scalars(h));
        const bool ok2 = h.ok != 0;
        const double temp_chi = ok2 ? h.chi : std::numeric_limits<double>::max();
        rho = (current_chi - temp_chi) / (h.scale + 1e-3);
        if (!ok2) rho = -1.0;  // the linear solver failed: the trial is rejected whatever its step looked like
        if (rho > 0 && std::isfinite(temp_chi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          current_chi = temp_chi;
        } else {
          lambda *= ni;
          ni *= 2;
          HIP_TRY(c, hipMemcpyAsync(d_poses, b + o_pose_bk, (size_t)NK * 56, hipMemcpyDeviceToDevice, st));  // pop()
          HIP_TRY(c, hipMemcpyAsync(d_points, b + o_pt_bk, (size_t)NP * 24, hipMemcpyDeviceToDevice, st));
          if (!std::isfinite(lambda)) break;
        }
        ++qmax;
      } while (rho < 0 && qmax < 10 && !stopped());
      if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) break;  // OptimizationAlgorithm::Terminate
    }
    return ORBFE_OK;
  };
  int32_t it1 = 0, it2 = 0;
  TRY(optimize(iters_first, it1));
  if (!stopped()) {
    // edge->chi2() is the chi2 of the last evaluated trial; isDepthPositive() reads the current estimates (Optimizer.cc:338-359)
    launch_ba_edges(st, E, d_poses, d_points, d_ek, d_ep, (const double*)(b + o_meas), b + o_st, (const double*)(b + o_info),
                    (const double*)(b + o_delta), prm, (double*)(b + o_err), (double*)(b + o_chi2), (double*)(b + o_rho), nullptr, nullptr,
                    b + o_depth);
    launch_lba_classify(st, E, (const double*)(b + o_last), b + o_depth, b + o_st, b + o_level, (double*)(b + o_info_eff),
                        (double*)(b + o_delta));
    TRY(optimize(iters_second, it2));
  }
  // final computeError() on every edge with the final estimates (Optimizer.cc:364-391)
  launch_ba_edges(st, E, d_poses, d_points, d_ek, d_ep, (const double*)(b + o_meas), b + o_st, (const double*)(b + o_info),
                  (const double*)(b + o_delta), prm, (double*)(b + o_err), (double*)(b + o_chi2), (double*)(b + o_rho), nullptr, nullptr,
                  b + o_depth);
  launch_lba_final(st, E, (const double*)(b + o_chi2), b + o_depth, b + o_st, b + o_bad);
  HIP_TRY(c, hipGetLastError());
  auto down = [&](void* dst, size_t o2, size_t bytes) -> hipError_t {
    return (bytes && dst) ? hipMemcpyAsync(dst, b + o2, bytes, hipMemcpyDeviceToHost, st) : hipSuccess;
  };
  HIP_TRY(c, down(o->poses, o_pose, (size_t)NK * 56));
  HIP_TRY(c, down(o->points, o_pt, (size_t)NP * 24));
  HIP_TRY(c, down(o->level, o_level, (size_t)E));
  HIP_TRY(c, down(o->chi2, o_chi2, (size_t)E * 8));
  HIP_TRY(c, down(o->bad, o_bad, (size_t)E));
  HIP_TRY(c, hipStreamSynchronize(st));
  if (o->iterations) {
    o->iterations[0] = it1;
    o->iterations[1] = it2;
  }
  return ORBFE_OK;
}


orbfe_status orbfe_pose_only_optimize(orbfe_ctx* c, int32_t n, const double* xw, const double* meas, const double* info, const float* sigma2,
                                      const double* pose_in, double fx, double fy, double cx, double cy, double bf, double* pose_out,
                                      uint8_t* inlier_out, int32_t* n_good) {
  ApiLock api_lk(c);
  if (!c || n < 0 || !pose_in || !pose_out || !n_good || (n && (!xw || !meas || !info || !sigma2)))
    return fail(c, ORBFE_EBADARG, "pose_only_optimize: NULL argument");
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t N = (size_t)std::max(n, 1);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  // inputs as ONE upload through the page-locked staging buffer, results as one download (five copies from pageable memory up and three down
  // were a fifth of the call)
  const size_t o_x = take(N * 24), o_m = take(N * 24), o_i = take(N * 8), o_s = take(N * 4), o_p = take(56), o_up_end = take(8),
               o_po = take(56), o_ng = take(8), o_in = take(N), o_out_end = take(8), o_e = take(N * 24), o_l = take(N), o_r = take(N);
  TRY(ensure_tmp(c, off));
  TRY(ensure_stage(c, std::max(o_up_end, o_out_end - o_po)));
  uint8_t* b = (uint8_t*)c->d_tmp;
  uint8_t* hs = c->main.h_stage;
  if (n) {
    std::memcpy(hs + o_x, xw, (size_t)n * 24);
    std::memcpy(hs + o_m, meas, (size_t)n * 24);
    std::memcpy(hs + o_i, info, (size_t)n * 8);
    std::memcpy(hs + o_s, sigma2, (size_t)n * 4);
  }
  std::memcpy(hs + o_p, pose_in, 56);
  HIP_TRY(c, hipMemcpyAsync(b, hs, o_up_end, hipMemcpyHostToDevice, c->stream));
  BaParamsDev prm = {fx, fy, cx, cy, bf};
  {
    StageTimer tm(c, ORBFE_STAGE_BA, c->stream);
    launch_pose_only(c->stream, n, (const double*)(b + o_x), (const double*)(b + o_m), (const double*)(b + o_i), (const float*)(b + o_s),
                     (const double*)(b + o_p), prm, (double)(float)std::sqrt(5.991), (double)(float)std::sqrt(7.815), (double*)(b + o_e),
                     b + o_l, b + o_r, b + o_in, (double*)(b + o_po), (int32_t*)(b + o_ng));
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(hs, b + o_po, (inlier_out && n ? o_in + (size_t)n : o_in) - o_po, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  std::memcpy(pose_out, hs, 56);
  std::memcpy(n_good, hs + (o_ng - o_po), 4);
  if (inlier_out && n) std::memcpy(inlier_out, hs + (o_in - o_po), (size_t)n);
  return ORBFE_OK;
}


}  // extern "C"
