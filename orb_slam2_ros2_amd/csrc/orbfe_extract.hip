// orbfe_extract.hip -- the launch sequence of an extraction (pyramid, FAST, quadtree, orientation, descriptors) and of the stereo match, and the
// entry points built on them: orbfe_extract* / orbfe_frame_stereo* / orbfe_frame_rgbd* / orbfe_stereo_match / orbfe_stereo_batch_device.
// (Split from orbfe_api.hip in r5, no change of behaviour.)
#include "orbfe_ctx.h"
// slots [s0, s0 + n) have just been (or are about to be) rewritten by an extraction; small: one that also built their row tables
void grid_invalidate(orbfe_ctx* c, int slot) {
  if (!c->grid_key) return;
  std::atomic<uint64_t>& a = c->grid_key[(size_t)slot];
  uint64_t v = a.load();
  while (!a.compare_exchange_weak(v, ((v >> 32) + 1) << 32)) {
  }
}
void note_slots_written(orbfe_ctx* c, int s0, int n, bool small) {
  for (int s = s0; s < s0 + n && s < c->cfg.max_images; ++s) grid_invalidate(c, s);
  if (!c->slot_table_ok) return;
  for (int s = s0; s < s0 + n && s < c->cfg.max_images; ++s) {
    c->slot_table_ok[(size_t)s] = small ? 1 : 0;
    if (small) c->pair_count_zero[(size_t)(s >> 1)] = 1;
  }
}
orbfe_status run_extract(orbfe_ctx* c, hipStream_t st, int img0, int n_img, hipEvent_t before_lists, bool timing, const ExtLevel0* ext,
                         const HostMirror* mirror) {
  // timing = false: a slot lane (orbfe_extract_slot) -- several of them run at once, so nothing shared by the context is touched:
  // no stage timers (their event lists belong to the main lane), no second stream
  // before_lists: event the keypoint-list / orientation / descriptor kernels must wait for (the previous batch's stereo match still
  // reads the arrays they rewrite); callers that do not pipeline have joined the stereo stream already
  const int nl = c->cfg.n_levels;
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  const size_t i0 = (size_t)img0;
  note_slots_written(c, img0, n_img, n_img <= 2);
  uint8_t* pyr = c->d_pyr + i0 * c->img_pitch;
  uint8_t* blur = c->d_blur + i0 * c->img_pitch;
  int32_t* n_cand = c->d_n_cand + i0 * nl;
  const bool overlap_blur = timing && c->blur_stream && c->prof != 1 && n_img >= 32;  // a frame or two: nothing to hide, only event latency to add (measured r3: the blur of one pair on the second stream, inside the captured graph: extract_batch 0.306 -> 0.366 ms)
  // the blur of LEVEL 0 needs nothing but the copy-in: it starts beside the resize (a third of the blur's work out of the way of the
  // moments, which are as memory-bound as it is and take the sum of the two times when they meet)
  const int l0_tiles = (overlap_blur && nl > 1) ? c->lv[1].bl_tile_base : 0;
  bool blur_queued = false;
  if (l0_tiles > 0 && !ext) {
    HIP_TRY(c, hipEventRecord(c->ev_blur_go, st));
    HIP_TRY(c, hipStreamWaitEvent(c->blur_stream, c->ev_blur_go, 0));
    launch_blur(c->blur_stream, c->d_lv, nl, 0, l0_tiles, pyr, blur, c->img_pitch, c->blur_taps, n_img);
  }
  // FAST's candidate counters are zeroed by the resize kernel (block 0): a memset between the blur and FAST is one more launch in the
  // chain -- 4.6 us of a 0.2 ms frame
  const bool zeroed_by_resize = !c->rs_regions.empty();  // (empty: the geometry rules the region-driven resize out -- the per-class tile launches)
  // A frame or two: FAST's candidate lists in shards (k_fast.hip, SH) -- where FAST is ONE launch of one-cell waves and the quadtree the
  // several-waves-per-tree launch that reads shards (the same conditions as launch_fast / launch_quadtree below)
  // (contexts of <= 16 slots never group levels: one tree per level, several waves each, when the launch has at most one tree per CU)
  const bool sharded = c->fast_shards > 1 && c->d_n_cand_sh && fast_single_launch(c->lv.data(), c->lvl_max_pw, c->lvl_max_ph, nl, n_img) &&
                       (long long)nl * n_img <= c->n_cu;
  int32_t* const n_cand_sh = sharded ? c->d_n_cand_sh + i0 * nl * c->fast_shards : nullptr;
  for (int i = 0; i < n_img; ++i) c->slot_sharded[(size_t)(i0 + i)] = sharded ? 1 : 0;
  {
    StageTimer t(c, ORBFE_STAGE_RESIZE, st, timing);
    if (zeroed_by_resize)
      launch_resize_regions(st, c->d_lv, nl, c->d_rs_regions, (int)c->rs_regions.size(), c->rg_tile_bytes, c->rg_xt_bytes, c->rg_yt_bytes,
                            c->d_rg_xtaps, c->d_rg_ytaps, pyr, c->img_pitch, n_img, ext ? ext->left : pyr + c->lv[0].plane_off,
                            ext ? ext->right : nullptr, ext ? ext->pitch : c->img_pitch, ext ? ext->stride : c->lv[0].stride,
                            ext ? ext->bytes : 0xFFFFFFFFu, ext ? 1 : 0, n_cand, n_img * nl, c->rg_pq, sharded ? n_cand_sh : c->d_qt_next + i0,
                            sharded ? n_img * nl * c->fast_shards : n_img);
    else
      launch_resize(st, c->d_lv, c->d_rs_tiles, c->rs_n, c->rs_bytes, c->d_taps, pyr, c->img_pitch, n_img);
  }
  if (ext) {  // the resize has also written level 0 of the pyramid (its blocks of the caller's images): the images are free, the level-0 blur may start
    if (ext->inputs_free) HIP_TRY(c, hipEventRecord(ext->inputs_free, st));
    if (l0_tiles > 0) {
      // ... and here, where the resize has just produced every level at once, the WHOLE blur goes to the second stream in one launch
      // (level 0 first): it runs beside FAST, mostly in the slots the eight launches leave at their tails, instead of the levels
      // above 0 waiting for FAST's last launch to drain (5.52 -> 5.50 ms per 512 pairs, four same-box rounds; two launches the same)
      HIP_TRY(c, hipEventRecord(c->ev_blur_go, st));
      HIP_TRY(c, hipStreamWaitEvent(c->blur_stream, c->ev_blur_go, 0));
      {
        StageTimer t(c, ORBFE_STAGE_BLUR, c->blur_stream);  // (events on the stream the kernel is launched on)
        if (c->mb_ok)  // the batches' blur on the integer matrix cores (k_blur_mfma.hip)
          launch_blur_mfma(c->blur_stream, c->d_lv, nl, c->mb, pyr, blur, c->img_pitch, c->d_mb_tx, c->d_mb_ty, n_img);
        else
          launch_blur(c->blur_stream, c->d_lv, nl, 0, c->bl_tiles, pyr, blur, c->img_pitch, c->blur_taps, n_img);
      }
      HIP_TRY(c, hipEventRecord(c->ev_blur_done, c->blur_stream));
      blur_queued = true;
    }
  }
  // Only the descriptors read the blurred planes, so the blur need not sit between resize and FAST: the levels above 0 are issued on a second
  // stream once FAST is done (beside FAST, which saturates the vector units, they cost more than they hide: +2 %) and runs UNDER the quadtree, which keeps 8 waves per CU busy with dependent LDS steps and leaves
  // the SIMDs idle (the blur uses no LDS, the quadtree all of it).  With stage timing on, or when several chunks share the
  // context, the blur stays in line.  (Measured and dropped: starting each level's quadtree under FAST of the smaller levels on
  // a third stream -- the tree waves then share their SIMDs with a VALU-saturating kernel and the dependent chain stretches:
  // 3.04 -> 4.6 ms per 128 pairs.)
  // A frame or two without stage timing: the blur's tiles ride in the quadtree launch below as extra workgroups (the launch has sixteen
  // tree workgroups per image on 256 CUs; the blurred planes are read by the descriptors only) -- one launch and its ~15 us off the chain.
  // With stage timing on the blur keeps its own launch so that the stages are timed apart.
  const bool qt_small = nl > 0 && (long long)nl * n_img <= c->n_cu;  // (= launch_quadtree's four-waves-per-tree condition below)
  const bool blur_in_qt = !overlap_blur && qt_small && c->prof == 0;
  if (!overlap_blur && !blur_in_qt) {
    StageTimer t(c, ORBFE_STAGE_BLUR, st, timing);
    if (c->mb_ok && n_img >= 32)  // (stage timing of a batch: the kernel the production schedule runs)
      launch_blur_mfma(st, c->d_lv, nl, c->mb, pyr, blur, c->img_pitch, c->d_mb_tx, c->d_mb_ty, n_img);
    else
      launch_blur(st, c->d_lv, nl, 0, c->bl_tiles, pyr, blur, c->img_pitch, c->blur_taps, n_img);
  }
  if (!zeroed_by_resize) {
    HIP_TRY(c, hipMemsetAsync(n_cand, 0, sizeof(int32_t) * (size_t)n_img * nl, st));
    if (sharded) HIP_TRY(c, hipMemsetAsync(n_cand_sh, 0, sizeof(int32_t) * (size_t)n_img * nl * c->fast_shards, st));
  }
  {
    StageTimer t(c, ORBFE_STAGE_FAST, st, timing);
    // r6: the launches of the SMALL levels -- fewer than 131072 cell x images: one or four cells per wave, each launch little more than a ramp
    // and a tail -- run on a stream of their own beside the launches of the large levels (same-box A/B, tools/exp/fast_side.sh: the step
    // -1.6 % on `rect`, -1.9 % on `camera`; levels 3-7 or alternating levels: less or nothing; the small levels merged into one launch: the gain halves; a THIRD stream for some of the
    // large levels: +8 %, whichever).  Only when there ARE large levels beside them.
    uint32_t side_mask = 0;
    if (overlap_blur && c->fast_stream && c->prof != 1) {
      if (c->fast_side_mask >= 0) {
        side_mask = (uint32_t)c->fast_side_mask & ((1u << nl) - 1u);
      } else {
        uint32_t small = 0, large = 0;
        for (int l = 0; l < nl; ++l) {
          if (c->lv[l].n_cells <= 0) continue;
          ((long long)c->lv[l].n_cells * n_img < 131072 ? small : large) |= 1u << l;
        }
        side_mask = large ? small : 0u;
      }
    }
    const bool side = side_mask != 0;
    if (side) {
      HIP_TRY(c, hipEventRecord(c->ev_fast_go, st));
      HIP_TRY(c, hipStreamWaitEvent(c->fast_stream, c->ev_fast_go, 0));
      launch_fast(c->fast_stream, c->d_lv, c->d_cells, c->lv.data(), c->lvl_max_pw, c->lvl_max_ph, pyr, c->img_pitch, c->cfg.fast_hi, c->cfg.fast_lo,
                  c->d_scr_a + i0 * c->scratch_pitch, c->scratch_pitch, n_cand, nl, n_img, c->fast_cpw, side_mask, c->fast_side_merge);
      HIP_TRY(c, hipEventRecord(c->ev_fast_side_done, c->fast_stream));
    }
    launch_fast(st, c->d_lv, c->d_cells, c->lv.data(), c->lvl_max_pw, c->lvl_max_ph, pyr, c->img_pitch, c->cfg.fast_hi, c->cfg.fast_lo,
                c->d_scr_a + i0 * c->scratch_pitch, c->scratch_pitch, n_cand, nl, n_img,
                c->fast_cpw, side ? ~side_mask : ~0u, false, n_cand_sh, sharded ? c->fast_shards : 1);
    if (side) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_fast_side_done, 0));
  }
  if (overlap_blur && !blur_queued) {
    HIP_TRY(c, hipEventRecord(c->ev_blur_go, st));
    HIP_TRY(c, hipStreamWaitEvent(c->blur_stream, c->ev_blur_go, 0));
    {
      StageTimer t(c, ORBFE_STAGE_BLUR, c->blur_stream);  // (events on the stream the kernel is launched on)
      launch_blur(c->blur_stream, c->d_lv, nl, l0_tiles, c->bl_tiles - l0_tiles, pyr, blur, c->img_pitch, c->blur_taps, n_img);
    }
    HIP_TRY(c, hipEventRecord(c->ev_blur_done, c->blur_stream));
  }
  {
    StageTimer t(c, ORBFE_STAGE_QUADTREE, st, timing);
    // LDS residency of the candidate records is traded against concurrency: the kernel is latency-bound (one wave per
    // tree, 40-150 dependent steps), so what matters most is that EVERY tree of the launch is resident at once; the
    // records go to LDS only as far as that still holds (measured at 1024 trees: 4 trees/CU 0.59 ms, 3 trees/CU 0.96 ms).
    // several levels per wave only where one wave per level would overfill the chip: waves per image so that the launch has about
    // sixteen tree waves per CU (one round) -- 4 waves per image from 1024 images, 2 from 2048, 1 from 4096 on 256 CUs
    int gsel = -1;  // -1: one wave per level
    if (nl > 4) {
      const long long per8 = (long long)c->n_cu * 16;  // sixteen tree waves per CU: 16-byte nodes, 128 VGPRs (k_quadtree.hip)
      if ((long long)n_img * 1 >= per8) gsel = 0;
      else if ((long long)n_img * 2 >= per8) gsel = 1;
      else if ((long long)n_img * 4 >= per8) gsel = 2;
    }
    const bool grouped = gsel >= 0;
    const int n_groups = gsel >= 0 ? std::min(nl, 1 << gsel) : nl;
    const QtGroups& qt_tab = gsel >= 0 ? c->qt_groups_of[gsel] : c->qt_single;
    const int trees = n_groups * n_img;
    const int per_cu = (trees + c->n_cu - 1) / c->n_cu;
    const size_t lds_cu = 160 * 1024 - 2048;
    const size_t node_bytes = quadtree_lds_bytes(c->node_cap, 0, c->sort_cap);
    size_t budget = lds_cu / (size_t)std::max(per_cu, 1);
    budget -= budget % 512;
    const int rec_cap = budget > node_bytes ? (int)std::min<size_t>((budget - node_bytes) / 4, (size_t)c->rec_cap) : 0;
    launch_quadtree(st, c->d_lv, nl, c->d_scr_a + i0 * c->scratch_pitch, c->d_scr_b + i0 * c->scratch_pitch,
                    c->d_scr_c + i0 * c->scratch_pitch, c->scratch_pitch, c->d_sel + i0 * NF, c->d_sel_count + i0 * nl,
                    c->cfg.n_features, n_cand, c->node_cap, c->sort_cap, rec_cap, n_img, 1, qt_tab, n_groups,
                    // helper waves for the data-parallel phases of a tree where the launch leaves the chip empty (a frame or two)
                    (!grouped && trees * 4 <= c->n_cu * 4) ? 4 : 1, c->d_qt_big ? c->d_qt_big + i0 * c->qt_big_pitch : nullptr,
                    c->qt_big_pitch, c->d_qt_tabs, blur_in_qt ? pyr : nullptr, blur, c->img_pitch, c->blur_taps, c->bl_tiles, c->d_qt_next + i0,
                    zeroed_by_resize && !sharded, n_cand_sh, sharded ? c->fast_shards : 1);
  }
  {
    StageTimer t(c, ORBFE_STAGE_BRIEF, st, timing);
    launch_orient_brief(st, c->d_lv, nl, pyr, blur, c->img_pitch, c->d_sel + i0 * NF, c->d_sel_count + i0 * nl, c->cfg.n_features,
                        c->d_pattern, c->umax, c->d_kps + i0 * NF, c->d_desc + i0 * NF * 32, c->d_aux + i0 * NF, c->d_n_kp + i0,
                        c->d_theta + i0 * NF, c->d_moments + i0 * NF, c->d_sincos + i0 * NF, c->d_kx + i0 * NF,
                        c->d_kpl + i0 * NF, c->cfg.height, n_img,
                        overlap_blur ? c->ev_blur_done : nullptr, before_lists, mirror ? mirror->kps : nullptr, mirror ? mirror->desc : nullptr,
                        mirror ? mirror->n_kp : nullptr, true, n_img <= 2 ? c->d_rowoff_slot : nullptr, c->d_rowlist_slot, c->d_n_match, c->cfg.height,
                        c->row_list_cap, img0, n_img <= 2 ? c->d_rt_flags : nullptr);
    // (the per-slot row tables: valid after an extraction of one or two images, stale after any other -- the flags are host state and
    //  this function also runs under graph CAPTURE, so the callers set them: extract_lane / note_slots_written)
  }
  HIP_TRY(c, hipGetLastError());
  return ORBFE_OK;
}

struct StereoHostOut {  // page-locked destinations for the results of one pair, written by k_stereo itself (nullable members)
  double *right_u, *depth;
  int32_t *best_right, *best_dist;
};
static orbfe_status run_stereo(orbfe_ctx* c, hipStream_t st, int slot_l0, int slot_r0, int slot_step, int pair0, int n_pairs, float fx,
                               float bf, const StereoHostOut* ho = nullptr, bool table_ready = false, bool timing = true) {
  // (c->d_pyr is read here, at launch time: a later swap of the pyramid buffers does not affect a launch already queued)
  // (the match counters are zeroed by k_rowtable)
  {
    StageTimer t(c, ORBFE_STAGE_STEREO, st, timing);  // (timing = false: a slot lane, which touches nothing the context shares)
    launch_stereo(st, c->d_lv, c->cfg.n_levels, c->d_pyr, c->img_pitch, c->d_kps, c->d_desc, c->d_aux, c->d_kx,
                  table_ready ? c->d_rowoff_slot : c->d_rowoff, table_ready ? c->d_rowlist_slot : c->d_rowlist,
                  c->cfg.height, c->row_list_cap, c->d_n_kp,
                  c->cfg.n_features, fx, bf,
                  c->cfg.width, kMeanThreshold, c->d_right_u, c->d_depth, c->d_n_match, c->d_best_right, c->d_best_dist, slot_l0,
                  slot_r0, slot_step, pair0, n_pairs, ho ? ho->right_u : nullptr, ho ? ho->depth : nullptr, ho ? ho->best_right : nullptr,
                  ho ? ho->best_dist : nullptr, table_ready, &c->st_rows);
  }
  HIP_TRY(c, hipGetLastError());
  return ORBFE_OK;
}

extern "C" {


// results of slots slot0..slot0+n_img-1 to the host through the lane's pinned staging buffer: one batch of D2H copies (full arrays:
// the counts are not known on the host yet), ONE synchronisation
static orbfe_status enqueue_fetch(orbfe_ctx* c, orbfe_ctx::Lane& ln, int slot0, int n_img, size_t o_kps, size_t o_desc, size_t o_cnt,
                                  bool want_kps, bool want_desc) {
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1), s0 = (size_t)slot0;
  HIP_TRY(c, hipMemcpyAsync(ln.h_stage + o_cnt, c->d_n_kp + s0, sizeof(int32_t) * n_img, hipMemcpyDeviceToHost, ln.stream));
  if (want_kps)
    HIP_TRY(c, hipMemcpyAsync(ln.h_stage + o_kps, c->d_kps + s0 * NF, (size_t)n_img * NF * sizeof(orbfe_keypoint), hipMemcpyDeviceToHost,
                              ln.stream));
  if (want_desc)
    HIP_TRY(c, hipMemcpyAsync(ln.h_stage + o_desc, c->d_desc + s0 * NF * 32, (size_t)n_img * NF * 32, hipMemcpyDeviceToHost, ln.stream));
  return ORBFE_OK;
}
static orbfe_status finish_fetch(orbfe_ctx* c, orbfe_ctx::Lane& ln, int n_img, size_t o_kps, size_t o_desc, size_t o_cnt,
                                 orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out, bool timing) {
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  HIP_TRY(c, hipStreamSynchronize(ln.stream));
  if (timing) drain_timers(c);
  const int32_t* cnt = (const int32_t*)(ln.h_stage + o_cnt);
  for (int i = 0; i < n_img; ++i) {
    const int32_t n = cnt[i];
    if (n < 0 || (size_t)n > NF) return fail(c, ORBFE_EDEVICE, "extract: corrupt count %d for image %d", n, i);
    if (kps) std::memcpy(kps + (size_t)i * NF, ln.h_stage + o_kps + (size_t)i * NF * sizeof(orbfe_keypoint), sizeof(orbfe_keypoint) * n);
    if (desc) std::memcpy(desc + (size_t)i * NF * 32, ln.h_stage + o_desc + (size_t)i * NF * 32, (size_t)32 * n);
    if (n_out) n_out[i] = n;
  }
  return ORBFE_OK;
}
static orbfe_status fetch_extract_results(orbfe_ctx* c, int n_img, size_t o_kps, size_t o_desc, size_t o_cnt, orbfe_keypoint* kps,
                                          uint8_t* desc, int32_t* n_out) {
  TRY(enqueue_fetch(c, c->main, 0, n_img, o_kps, o_desc, o_cnt, kps != nullptr, desc != nullptr));
  return finish_fetch(c, c->main, n_img, o_kps, o_desc, o_cnt, kps, desc, n_out, true);
}

// Host images -> slots [slot0, slot0 + n_img) on lane `ln`: copy-in, the launch sequence, results back, one synchronisation.  One or
// two images (the drop-in call shape) are launch-bound: the whole sequence is captured once per lane into a hipGraph and replayed.
// fs (orbfe_frame_stereo; two images): the stereo match of (slot0, slot0 + 1) follows the extraction in the same launch sequence, its
// results come back through the staging buffer as the features do
struct FrameStereoReq {
  float fx, bf;
  double *right_u, *depth;  // [n_features], caller's
  int32_t* n_matches;
};
// fr (orbfe_frame_rgbd_image; one image): the image may be a 3-channel one (converted to gray on the way into level 0), and the RGB-D tail of
// the Frame constructor -- undistortion, depth / rightU lookup -- follows the extraction in the same launch sequence; the depth image is
// read by that kernel straight from the staging buffer (one 2- or 4-byte read per keypoint: it is never uploaded)
struct FrameRgbdReq {
  int32_t color_order;  // 0: the image is gray | 1: RGB | 2: BGR
  orbfe_camera cam;
  const void* depth;    // nullable: undistortion only
  int32_t depth_type;
  size_t depth_stride;
  float depth_scale;
  double *depth_out, *right_u_out;  // [n_features], caller's, nullable
  bool no_tail;                     // orbfe_extract_color: the conversion and the extraction only (the keypoints stay as extracted)
};
struct FrameRgbdKey {  // what of a request is baked into a captured launch sequence
  int32_t color_order, has_depth /* 2: no tail at all */, depth_type;
  size_t depth_stride;
  float depth_scale;
  orbfe_camera cam;
};
static orbfe_status extract_lane(orbfe_ctx* c, orbfe_ctx::Lane& ln, int slot0, int n_img, const uint8_t* const* imgs, size_t stride,
                                 orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out, bool timing, const FrameStereoReq* fs = nullptr,
                                 const FrameRgbdReq* fr = nullptr, int phase = 0) {
  // phase 0: the whole call | 1: stage + enqueue, then return (orbfe_extract_slot_begin: kps / desc non-NULL markers, nothing is written
  // through them) | 2: wait + deliver (orbfe_extract_slot_end; imgs is not looked at)
  const LevelDev& L0 = c->lv[0];
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  const bool color = fr && fr->color_order != 0;
  const size_t crow = align_up((size_t)c->cfg.width * 3, 16) + 16;  // staged colour rows: 4-aligned, with room for the last 12-byte group
  const size_t plane = align_up(color ? crow * (size_t)L0.h : (size_t)L0.stride * L0.h, 256);
  const size_t o_kps = (size_t)n_img * plane, o_desc = o_kps + align_up((size_t)n_img * NF * sizeof(orbfe_keypoint), 256);
  const size_t o_cnt = o_desc + align_up((size_t)n_img * NF * 32, 256), o_ru = o_cnt + align_up((size_t)n_img * 4, 256);
  const bool extra = fs || fr;
  const size_t o_dp = o_ru + (extra ? align_up(NF * 8, 256) : 0), o_dimg = o_dp + (extra ? align_up(NF * 8, 256) : 0);
  const size_t d_bytes = (fr && fr->depth) ? fr->depth_stride * (size_t)c->cfg.height : 0;
  const size_t total = o_dimg + align_up(d_bytes, 256);
  TRY(ensure_stage(c, ln, total));
  for (int i = 0; i < n_img && phase != 2; ++i) {
    if (!imgs[i]) return fail(c, ORBFE_EBADARG, "extract: image %d is NULL", i);
    uint8_t* dst = ln.h_stage + (size_t)i * plane;
    if (color)
      for (int y = 0; y < L0.h; ++y) std::memcpy(dst + (size_t)y * crow, imgs[i] + (size_t)y * stride, (size_t)c->cfg.width * 3);
    else
      for (int y = 0; y < L0.h; ++y) std::memcpy(dst + (size_t)y * L0.stride, imgs[i] + (size_t)y * stride, (size_t)c->cfg.width);
  }
  if (d_bytes) std::memcpy(ln.h_stage + o_dimg, fr->depth, d_bytes);
  FrameRgbdKey rkey;
  static_assert(sizeof(FrameRgbdKey) <= sizeof(orbfe_ctx::GraphEntry::rkey), "GraphEntry::rkey");
  std::memset(&rkey, 0, sizeof rkey);
  if (fr) {
    rkey.color_order = fr->color_order, rkey.has_depth = fr->no_tail ? 2 : (fr->depth ? 1 : 0), rkey.depth_type = fr->depth_type;
    rkey.depth_stride = fr->depth_stride, rkey.depth_scale = fr->depth_scale, rkey.cam = fr->cam;
  }
  uint8_t* const pyr_now = c->d_pyr;
  if (phase != 2) note_slots_written(c, slot0, n_img, n_img <= 2);  // (also when a captured graph is replayed: run_extract does not run then)
  // The results come back through the staging buffer too: the orientation and the descriptor kernels write keypoints, counts and
  // descriptors there themselves (posted PCIe writes, ~120 KB per image) beside the device arrays the stereo match reads -- three
  // device-to-host copies queued behind the last kernel cost ~17 us of a ~0.3 ms call.  More than two images: the copies.
  const bool mirror_on = n_img <= 2;
  const bool rgbd_tail = fr && !fr->no_tail;
  HostMirror mir = {(kps && !rgbd_tail) ? (orbfe_keypoint*)(ln.h_stage + o_kps) : nullptr, desc ? ln.h_stage + o_desc : nullptr, (int32_t*)(ln.h_stage + o_cnt)};
  auto enqueue_all = [&]() -> orbfe_status {
    // (more images: both in ONE copy -- rows = images: the staging planes are `plane` bytes apart, the pyramid slots img_pitch)
    // One or two images: level 0 is read from the page-locked staging planes by a copy KERNEL (16 bytes per load over PCIe, every byte
    // once) -- 6 us less per pair than the copy engine's 27 us transfer and its hand-over to the compute queue (same box, alternating:
    // extraction 0.278 -> 0.271 ms).  (The resize reading the staged planes itself was measured in r3 and dropped: it reads a pixel more than once.)
    if (color)  // (one image) cv::cvtColor of Tracking::grabFrame on the way in: the kernel reads the staged rows itself
      launch_cvt_gray(ln.stream, ln.h_stage, crow, pyr_now + (size_t)slot0 * c->img_pitch + L0.plane_off, L0.stride, c->cfg.width, c->cfg.height,
                      fr->color_order, c->cfg.gray_variant ? 1 : 0);
    else if (n_img <= 2)
      launch_load_level0(ln.stream, ln.h_stage, nullptr, (size_t)L0.stride, plane, pyr_now, c->img_pitch, (uint32_t)L0.plane_off, L0.stride, c->cfg.width,
                         L0.h, slot0, 1, n_img);
    else
      HIP_TRY(c, hipMemcpy2DAsync(pyr_now + (size_t)slot0 * c->img_pitch + L0.plane_off, c->img_pitch, ln.h_stage, plane, (size_t)L0.stride * L0.h,
                                (size_t)n_img, hipMemcpyHostToDevice, ln.stream));
    TRY(run_extract(c, ln.stream, slot0, n_img, nullptr, timing, nullptr, mirror_on ? &mir : nullptr));
    if (fs) {
      // the right image's row table and the zeroed pair counter come out of the extraction above when this context builds them there
      const StereoHostOut ho = {(double*)(ln.h_stage + o_ru), (double*)(ln.h_stage + o_dp), nullptr, nullptr};
      const bool table_ready = c->slot_table_ok && c->slot_table_ok[(size_t)slot0 + 1] != 0;
      TRY(run_stereo(c, ln.stream, slot0, slot0 + 1, 0, slot0 / 2, 1, fs->fx, fs->bf, &ho, table_ready, timing));
    }
    if (rgbd_tail) {
      launch_frame_rgbd(ln.stream, c->d_kps + (size_t)slot0 * NF, c->d_n_kp + slot0, (int)NF, fr->cam, d_bytes ? ln.h_stage + o_dimg : nullptr,
                        fr->depth_type, fr->depth_stride, fr->depth_scale, (double*)(ln.h_stage + o_dp), (double*)(ln.h_stage + o_ru),
                        (orbfe_keypoint*)(ln.h_stage + o_kps));
      HIP_TRY(c, hipGetLastError());
    }
    if (mirror_on) return ORBFE_OK;
    return enqueue_fetch(c, ln, slot0, n_img, o_kps, o_desc, o_cnt, kps != nullptr, desc != nullptr);
  };
  // the match has counted into the pair's counter: a later orbfe_stereo_match on these slots clears it first
  auto finish = [&]() -> orbfe_status {
    TRY(finish_fetch(c, ln, n_img, o_kps, o_desc, o_cnt, kps, desc, n_out, timing));
    if (fs) {
      if (c->pair_count_zero) c->pair_count_zero[(size_t)(slot0 / 2)] = 0;
      const double* ru = (const double*)(ln.h_stage + o_ru);
      const double* dp = (const double*)(ln.h_stage + o_dp);
      const size_t n = (size_t)c->cfg.n_features;
      int32_t nm = 0;  // k_stereo counts exactly the features it gives a right coordinate (>= 0; -1 otherwise)
      for (size_t i = 0; i < n; ++i) nm += ru[i] >= 0.0 ? 1 : 0;
      if (fs->right_u && n) std::memcpy(fs->right_u, ru, sizeof(double) * n);
      if (fs->depth && n) std::memcpy(fs->depth, dp, sizeof(double) * n);
      if (fs->n_matches) *fs->n_matches = nm;
    }
    if (rgbd_tail) {
      const size_t n = (size_t)c->cfg.n_features;
      if (fr->depth_out && n) std::memcpy(fr->depth_out, ln.h_stage + o_dp, sizeof(double) * n);
      if (fr->right_u_out && n) std::memcpy(fr->right_u_out, ln.h_stage + o_ru, sizeof(double) * n);
    }
    return ORBFE_OK;
  };
  if (phase == 2) return finish();
  if (c->use_graphs && ln.use_graphs && c->prof == 0 && n_img <= 2) {
    hipGraphExec_t exec = nullptr;
    for (auto it = ln.graphs.begin(); it != ln.graphs.end();) {
      if (it->stage != ln.h_stage) {  // the staging buffer was re-allocated: the captured addresses are stale
        (void)hipGraphExecDestroy(it->exec);
        it = ln.graphs.erase(it);
        continue;
      }
      if (it->slot0 == slot0 && it->n_img == n_img && it->want_kps == (kps != nullptr) && it->want_desc == (desc != nullptr) && it->pyr == pyr_now &&
          it->stereo == (fs != nullptr) && (!fs || (it->fx == fs->fx && it->bf == fs->bf)) && it->rgbd == (fr != nullptr) &&
          (!fr || std::memcmp(&it->rkey, &rkey, sizeof rkey) == 0))
        exec = it->exec;
      ++it;
    }
    if (!exec) {
      hipGraph_t g = nullptr;
      bool ok = hipStreamBeginCapture(ln.stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
      const orbfe_status st = ok ? enqueue_all() : ORBFE_EDEVICE;
      if (ok) ok = hipStreamEndCapture(ln.stream, &g) == hipSuccess && st == ORBFE_OK && g;
      if (ok) ok = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) == hipSuccess;
      if (g) (void)hipGraphDestroy(g);
      if (ok) {
        orbfe_ctx::GraphEntry ge{slot0, n_img, kps != nullptr, desc != nullptr, fs != nullptr, fs ? fs->fx : 0.f, fs ? fs->bf : 0.f, ln.h_stage, pyr_now, exec};
        ge.rgbd = fr != nullptr;
        std::memcpy(ge.rkey, &rkey, sizeof rkey);
        ln.graphs.push_back(ge);
      } else {
        (void)hipGetLastError();
        exec = nullptr;
        ln.use_graphs = false;  // this runtime cannot capture the sequence: plain launches on this lane from now on
      }
    }
    if (exec) {
      HIP_TRY(c, hipGraphLaunch(exec, ln.stream));
      return phase == 1 ? ORBFE_OK : finish();
    }
  }
  TRY(enqueue_all());
  return phase == 1 ? ORBFE_OK : finish();
}

orbfe_status orbfe_extract_batch(orbfe_ctx* c, int32_t n_img, const uint8_t* const* imgs, size_t stride, orbfe_keypoint* kps,
                                 uint8_t* desc, int32_t* n_out) {
  ApiLock api_lk(c);
  if (!c || !imgs || n_img < 0) return fail(c, ORBFE_EBADARG, "extract_batch: NULL argument");
  if (n_img > c->cfg.max_images) return fail(c, ORBFE_ECAPACITY, "extract_batch: %d images > max_images %d", n_img, c->cfg.max_images);
  if (stride < (size_t)c->cfg.width) return fail(c, ORBFE_EBADARG, "extract_batch: stride %zu < width %d", stride, c->cfg.width);
  if (n_img == 0) return ORBFE_OK;
  TRY(slots_idle(c, 0, n_img, "extract_batch"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  return extract_lane(c, c->main, 0, n_img, imgs, stride, kps, desc, n_out, true);
}

// The device work of Frame::createStereo (include/ORB_SLAM2/Frame.h:313-323: the constructor's two extractions, src/Frame.cc:100-105,
// then ORBMatcher::searchByStereo) as ONE call: both images up, the extraction of slots 0 and 1 and their stereo match as one launch sequence (one graph replay), every
// result back through the staging buffer, one synchronisation.  Same results as orbfe_extract_batch([left, right]) followed by
// orbfe_stereo_match(0, 1) -- the same kernels in the same order -- without the second call's launch, copy and wake-up.
orbfe_status orbfe_frame_stereo(orbfe_ctx* c, const uint8_t* left, const uint8_t* right, size_t stride, float fx, float bf,
                                orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out, double* right_u, double* depth, int32_t* n_matches) {
  ApiLock api_lk(c);
  if (!c || !left || !right) return fail(c, ORBFE_EBADARG, "frame_stereo: NULL argument");
  if (c->cfg.max_images < 2) return fail(c, ORBFE_ECAPACITY, "frame_stereo: the context holds %d image(s), a stereo frame needs 2", c->cfg.max_images);
  if (stride < (size_t)c->cfg.width) return fail(c, ORBFE_EBADARG, "frame_stereo: stride %zu < width %d", stride, c->cfg.width);
  TRY(slots_idle(c, 0, 2, "frame_stereo"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const uint8_t* imgs[2] = {left, right};
  const FrameStereoReq fs = {fx, bf, right_u, depth, n_matches};
  return extract_lane(c, c->main, 0, 2, imgs, stride, kps, desc, n_out, true, &fs);
}

// One image -> slot `slot` on that slot's own lane.  Calls on DIFFERENT slots may run at the same time on different threads.
orbfe_status orbfe_extract_slot(orbfe_ctx* c, int32_t slot, const uint8_t* img, size_t stride, orbfe_keypoint* kps, uint8_t* desc,
                                int32_t* n_out) {
  const uint8_t* one[1] = {img};
  return orbfe_extract_slots(c, slot, 1, one, stride, kps, desc, n_out);
}

// n_img images -> slots [slot0, slot0 + n_img) on slot0's lane: what a caller does who holds BOTH images of a stereo frame when the
// first extract() is reached (host/orbfe_shim.hpp keeps one orbfe_extract_slot per extract() thread: pairing the threads was measured and dropped)
static orbfe_status extract_slots_impl(orbfe_ctx* c, int32_t slot, int32_t n_img, const uint8_t* const* imgs, size_t stride, orbfe_keypoint* kps,
                                       uint8_t* desc, int32_t* n_out, const FrameStereoReq* fs, const FrameRgbdReq* fr = nullptr, int phase = 0) {
  if (!c || !imgs || n_img < 1) return fail(c, ORBFE_EBADARG, "extract_slots: NULL argument / no image");
  const uint8_t* img = imgs[0];
  if (!img && phase != 2) return fail(c, ORBFE_EBADARG, "extract_slot: NULL argument");
  if (slot < 0 || slot + n_img > c->cfg.max_images) return fail(c, ORBFE_EBADARG, "extract_slot: slots %d..%d of %d", slot, slot + n_img - 1, c->cfg.max_images);
  if (phase != 2 && stride < (size_t)c->cfg.width * ((fr && fr->color_order) ? 3 : 1)) return fail(c, ORBFE_EBADARG, "extract_slot: stride %zu < the row's %d bytes", stride, c->cfg.width * ((fr && fr->color_order) ? 3 : 1));
  HIP_TRY(c, hipSetDevice(c->device));
  // the lanes of EVERY slot the call writes, created on first use and locked in index order (a concurrent slot call on any of them waits;
  // two multi-slot calls cannot deadlock); the work runs on the first slot's lane
  std::vector<orbfe_ctx::Lane*> lanes((size_t)n_img, nullptr);
  {
    std::lock_guard<std::mutex> lk(c->slot_lane_mu);
    for (int k = 0; k < n_img; ++k) {
      if (!c->slot_lane[(size_t)(slot + k)]) {
        std::unique_ptr<orbfe_ctx::Lane> fresh(new orbfe_ctx::Lane());
        HIP_TRY(c, hipStreamCreateWithFlags(&fresh->stream, hipStreamNonBlocking));
        fresh->own_stream = true;
        if (hipEventCreateWithFlags(&fresh->ev_main, hipEventDisableTiming) != hipSuccess) {
          (void)hipStreamDestroy(fresh->stream);
          return fail(c, ORBFE_EDEVICE, "extract_slot: cannot create the lane event");
        }
        c->slot_lane[(size_t)(slot + k)] = std::move(fresh);
      }
      lanes[(size_t)k] = c->slot_lane[(size_t)(slot + k)].get();
    }
  }
  std::vector<std::unique_lock<std::mutex>> held;
  held.reserve(lanes.size());
  for (orbfe_ctx::Lane* l : lanes) held.emplace_back(l->mu);
  orbfe_ctx::Lane* ln = lanes[0];
  for (orbfe_ctx::Lane* l : lanes)
    if (l->pending != (phase == 2)) return fail(c, ORBFE_EBADARG, phase == 2 ? "extract_slot_end: no orbfe_extract_slot_begin is outstanding on slot %d"
                                                                               : "extract_slot: slot %d has an outstanding orbfe_extract_slot_begin", slot);
  if (phase == 2) {
    ln->pending = false;
    return extract_lane(c, *ln, slot, n_img, imgs, stride, kps, desc, n_out, false, nullptr, nullptr, 2);
  }
  // a stereo match of an earlier device batch may still be reading the slot arrays (the flag is only read here: the calls that
  // change it must not overlap with slot calls)
  if (c->stereo_pending) HIP_TRY(c, hipStreamWaitEvent(ln->stream, c->ev_stereo_done, 0));
  // ... and an asynchronous batch call may have left work on the context stream that still writes this slot.  The marker is recorded
  // under the API lock: another thread may be inside hipStreamBeginCapture on the context stream (the first orbfe_extract /
  // orbfe_extract_batch of a shape), and an event recorded into that capture would pull this lane's stream into it -- both graphs fail
  {
    ApiLock api_lk(c);
    HIP_TRY(c, hipEventRecord(ln->ev_main, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(ln->stream, ln->ev_main, 0));
  }
  const orbfe_status st = extract_lane(c, *ln, slot, n_img, imgs, stride, kps, desc, n_out, false, fs, fr, phase);
  if (phase == 1 && st == ORBFE_OK) ln->pending = true;
  return st;
}
orbfe_status orbfe_extract_slot_begin(orbfe_ctx* c, int32_t slot, const uint8_t* img, size_t stride) {
  const uint8_t* one[1] = {img};
  static orbfe_keypoint kp_marker;  // (non-NULL: the launch sequence mirrors keypoints and descriptors into the staging buffer; nothing is written here)
  static uint8_t desc_marker;
  return extract_slots_impl(c, slot, 1, one, stride, &kp_marker, &desc_marker, nullptr, nullptr, nullptr, 1);
}
orbfe_status orbfe_extract_slot_end(orbfe_ctx* c, int32_t slot, orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out) {
  const uint8_t* one[1] = {nullptr};
  return extract_slots_impl(c, slot, 1, one, 0, kps, desc, n_out, nullptr, nullptr, 2);
}

orbfe_status orbfe_extract_slots(orbfe_ctx* c, int32_t slot, int32_t n_img, const uint8_t* const* imgs, size_t stride, orbfe_keypoint* kps,
                                 uint8_t* desc, int32_t* n_out) {
  return extract_slots_impl(c, slot, n_img, imgs, stride, kps, desc, n_out, nullptr);
}
// The device work of Frame::createRGBD (include/ORB_SLAM2/Frame.h:326-331) for one image as ONE call: Tracking::grabFrame's cvtColor when the
// image has three channels (src/Tracking.cc:55-68), the extraction (the RGB-D Frame constructor, src/Frame.cc:125-135), then
// Camera::undistortPoints and the depth / rightU lookup (:136-158) -- what orbfe_extract_color / orbfe_extract_slot followed by
// orbfe_frame_rgbd do in two calls.  On the slot's own lane (as orbfe_extract_slot).  The depth image is never uploaded: the last kernel
// reads one value per keypoint from the page-locked staging copy.
orbfe_status orbfe_frame_rgbd_image(orbfe_ctx* c, int32_t slot, const uint8_t* img, size_t stride, int32_t color_order, const orbfe_camera* cam,
                                    const void* depth, int32_t depth_type, size_t depth_stride, float depth_scale, orbfe_keypoint* kps_undistorted,
                                    uint8_t* desc, int32_t* n_out, double* depth_out, double* right_u_out) {
  if (!c || !img || !cam) return fail(c, ORBFE_EBADARG, "frame_rgbd_image: NULL argument");
  if (color_order < 0 || color_order > 2) return fail(c, ORBFE_EBADARG, "frame_rgbd_image: color_order %d (0 = gray, 1 = RGB, 2 = BGR)", color_order);
  const size_t px = depth_type == 0 ? 2 : 4;
  if (depth && (depth_type < 0 || depth_type > 1 || depth_stride < (size_t)c->cfg.width * px || !(depth_scale > 0)))
    return fail(c, ORBFE_EBADARG, "frame_rgbd_image: depth type %d stride %zu scale %g", depth_type, depth_stride, (double)depth_scale);
  const uint8_t* one[1] = {img};
  const FrameRgbdReq fr = {color_order, *cam, depth, depth_type, depth_stride, depth_scale, depth_out, right_u_out, false};
  return extract_slots_impl(c, slot, 1, one, stride, kps_undistorted, desc, n_out, nullptr, &fr);
}
// orbfe_frame_stereo into the slot pair (slot_left, slot_left + 1), slot_left even, on slot_left's lane: what the drop-in's frame-level
// adapter calls (the extractor objects rotate over the context's slots; a Frame's device-side features live as long as its slots do)
orbfe_status orbfe_frame_stereo_slots(orbfe_ctx* c, int32_t slot_left, const uint8_t* left, const uint8_t* right, size_t stride, float fx,
                                      float bf, orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out, double* right_u, double* depth,
                                      int32_t* n_matches) {
  if (!c || !left || !right) return fail(c, ORBFE_EBADARG, "frame_stereo_slots: NULL argument");
  if (slot_left < 0 || (slot_left & 1) || slot_left + 2 > c->cfg.max_images)
    return fail(c, ORBFE_EBADARG, "frame_stereo_slots: slot %d (even, and slot + 1 < max_images %d)", slot_left, c->cfg.max_images);
  const uint8_t* imgs[2] = {left, right};
  const FrameStereoReq fs = {fx, bf, right_u, depth, n_matches};
  return extract_slots_impl(c, slot_left, 2, imgs, stride, kps, desc, n_out, &fs);
}

orbfe_status orbfe_extract(orbfe_ctx* c, const uint8_t* img, size_t stride, orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out) {
  const uint8_t* one[1] = {img};
  return orbfe_extract_batch(c, 1, one, stride, kps, desc, n_out);
}

orbfe_status orbfe_extract_color(orbfe_ctx* c, const uint8_t* img, size_t stride, int32_t color_order, orbfe_keypoint* kps, uint8_t* desc,
                                 int32_t* n_out) {
  // (through the one-frame launch sequence since late r4: the conversion kernel reads the staged colour rows itself, the sequence is
  //  replayed from a captured graph and the results come back through the staging buffer -- the same path as orbfe_extract_batch)
  ApiLock api_lk(c);
  if (!c || !img) return fail(c, ORBFE_EBADARG, "extract_color: NULL argument");
  if (color_order != 1 && color_order != 2) return fail(c, ORBFE_EBADARG, "extract_color: color_order %d (1 = RGB, 2 = BGR)", color_order);
  if (stride < (size_t)c->cfg.width * 3) return fail(c, ORBFE_EBADARG, "extract_color: stride %zu < 3 * width", stride);
  TRY(slots_idle(c, 0, 1, "extract_color"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const uint8_t* one[1] = {img};
  FrameRgbdReq fr;
  std::memset(&fr, 0, sizeof fr);
  fr.color_order = color_order, fr.no_tail = true;
  return extract_lane(c, c->main, 0, 1, one, stride, kps, desc, n_out, true, nullptr, &fr);
}

orbfe_status orbfe_frame_rgbd(orbfe_ctx* c, int32_t slot, const orbfe_camera* cam, const void* depth, int32_t depth_type,
                              size_t depth_stride, float depth_scale, orbfe_keypoint* kps_out, double* depth_out, double* right_u_out) {
  ApiLock api_lk(c);
  if (!c || !cam || slot < 0 || slot >= c->cfg.max_images) return fail(c, ORBFE_EBADARG, "frame_rgbd: bad slot / NULL camera");
  const size_t px = depth_type == 0 ? 2 : 4;
  if (depth && (depth_type < 0 || depth_type > 1 || depth_stride < (size_t)c->cfg.width * px || !(depth_scale > 0)))
    return fail(c, ORBFE_EBADARG, "frame_rgbd: depth type %d stride %zu scale %g", depth_type, depth_stride, (double)depth_scale);
  TRY(slots_idle(c, slot, 1, "frame_rgbd"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  const size_t d_bytes = depth ? depth_stride * (size_t)c->cfg.height : 0;
  // The depth image is staged in page-locked memory and READ FROM THERE by the kernel (one 2- or 4-byte value per keypoint: uploading
  // 614 KB of a 640 x 480 16-bit image for 1000 reads was a third of the call); depth, rightU and the undistorted keypoints are written
  // to the staging buffer by the kernel as well: one 4-byte copy (the count), one synchronisation.
  const size_t h_res = align_up(d_bytes, 256), h_ru = h_res + align_up(NF * 8, 256), h_n = h_ru + align_up(NF * 8, 256), h_k = h_n + 256,
               h_total = h_k + align_up(NF * sizeof(orbfe_keypoint), 256);
  TRY(ensure_stage(c, h_total));
  uint8_t* hs = c->main.h_stage;
  if (depth) std::memcpy(hs, depth, d_bytes);
  grid_invalidate(c, slot);  // (the keypoints move: a grid kept for the slot is stale)
  launch_frame_rgbd(c->stream, c->d_kps + (size_t)slot * NF, c->d_n_kp + slot, (int)NF, *cam, depth ? hs : nullptr, depth_type, depth_stride,
                    depth_scale, (double*)(hs + h_res), (double*)(hs + h_ru), kps_out ? (orbfe_keypoint*)(hs + h_k) : nullptr);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(hs + h_n, c->d_n_kp + slot, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int32_t n = 0;
  std::memcpy(&n, hs + h_n, 4);
  if (depth_out) std::memcpy(depth_out, hs + h_res, NF * 8);
  if (right_u_out) std::memcpy(right_u_out, hs + h_ru, NF * 8);
  if (kps_out && n > 0) std::memcpy(kps_out, hs + h_k, sizeof(orbfe_keypoint) * (size_t)std::min<int64_t>(n, (int64_t)NF));
  return ORBFE_OK;
}

orbfe_status orbfe_get_pyramid(orbfe_ctx* c, int32_t slot, int32_t level, int32_t blurred, uint8_t* dst) {
  ApiLock api_lk(c);
  if (!c || !dst || slot < 0 || slot >= c->cfg.max_images || level < 0 || level >= c->cfg.n_levels)
    return fail(c, ORBFE_EBADARG, "get_pyramid: slot %d level %d", slot, level);
  TRY(slots_idle(c, slot, 1, "get_pyramid"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const LevelDev& L = c->lv[level];
  const uint8_t* src = (blurred ? c->d_blur : c->d_pyr) + (size_t)slot * c->img_pitch + L.plane_off;
  HIP_TRY(c, hipMemcpy2DAsync(dst, L.w, src, L.stride, L.w, L.h, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ORBFE_OK;
}

orbfe_status orbfe_stereo_match(orbfe_ctx* c, int32_t slot_left, int32_t slot_right, float fx, float bf, double* right_u,
                                double* depth, int32_t* n_matches, int32_t* best_right, int32_t* best_dist) {
  ApiLock api_lk(c);
  if (!c || slot_left < 0 || slot_right < 0 || slot_left >= c->cfg.max_images || slot_right >= c->cfg.max_images)
    return fail(c, ORBFE_EBADARG, "stereo_match: slots %d/%d", slot_left, slot_right);
  TRY(slots_idle(c, slot_left, 1, "stereo_match"));
  TRY(slots_idle(c, slot_right, 1, "stereo_match"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const int pair = slot_left / 2;
  // the kernel writes the requested arrays into the page-locked staging buffer itself; only the match count is copied (4 bytes)
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  const size_t o_ru = 0, o_dp = align_up(NF * 8, 256), o_br = o_dp + align_up(NF * 8, 256), o_bd = o_br + align_up(NF * 4, 256),
               o_nm = o_bd + align_up(NF * 4, 256), total = o_nm + 256;
  TRY(ensure_stage(c, total));
  uint8_t* h = c->main.h_stage;
  const StereoHostOut ho = {right_u ? (double*)(h + o_ru) : nullptr, depth ? (double*)(h + o_dp) : nullptr,
                            best_right ? (int32_t*)(h + o_br) : nullptr, best_dist ? (int32_t*)(h + o_bd) : nullptr};
  // The right image's row table: built by its extraction when that was a one- or two-image call of this (small) context -- the match
  // is then k_stereo alone; the pair's counter was zeroed there too unless an earlier match has counted into it since.
  const bool table_ready = c->slot_table_ok && c->slot_table_ok[(size_t)slot_right] != 0;
  if (table_ready && !c->pair_count_zero[(size_t)pair].exchange(0)) HIP_TRY(c, hipMemsetAsync(c->d_n_match + pair, 0, sizeof(int32_t), c->stream));
  TRY(run_stereo(c, c->stream, slot_left, slot_right, 0, pair, 1, fx, bf, &ho, table_ready));
  // (the count: with right_u in the staging buffer it is counted there -- k_stereo counts exactly the features it gives a right coordinate --
  //  and the 4-byte copy, a transfer of its own behind the kernel, is left out)
  const bool count_on_host = right_u != nullptr;
  if (n_matches && !count_on_host) HIP_TRY(c, hipMemcpyAsync(h + o_nm, c->d_n_match + pair, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  const size_t n = (size_t)c->cfg.n_features;
  if (right_u && n) std::memcpy(right_u, h + o_ru, sizeof(double) * n);
  if (depth && n) std::memcpy(depth, h + o_dp, sizeof(double) * n);
  if (best_right && n) std::memcpy(best_right, h + o_br, sizeof(int32_t) * n);
  if (best_dist && n) std::memcpy(best_dist, h + o_bd, sizeof(int32_t) * n);
  if (n_matches) {
    if (count_on_host) {
      int32_t nm = 0;
      const double* ru = (const double*)(h + o_ru);
      for (size_t i = 0; i < n; ++i) nm += ru[i] >= 0.0 ? 1 : 0;
      *n_matches = nm;
    } else
      std::memcpy(n_matches, h + o_nm, sizeof(int32_t));
  }
  return ORBFE_OK;
}

}  // extern "C"

orbfe_status batch_device_core(orbfe_ctx* c, const uint8_t* d_left, const uint8_t* d_right, size_t stride, size_t image_pitch,
                                      int32_t n_pairs, float fx, float bf, const PackDst* pack) {
  const LevelDev& L0 = c->lv[0];
  // The stereo match goes to its own stream and this call returns with it still queued; the next call starts its copy-in / resize / FAST
  // on the context stream right away, into the OTHER pyramid buffer, and only its keypoint-list kernels wait for the match (they rewrite
  // what it reads).  Every other entry point joins the stereo stream first.  With stage timing on (prof == 1) everything runs in line.
  const bool pipe = c->pipeline_stereo && c->stereo_stream && c->prof != 1 && n_pairs >= 16;
  if (pipe) {
    if (!c->d_pyr_alt) {
      // (cleared ON THE CONTEXT STREAM: a null-stream hipMemset returns before the device has run it and is not ordered with this
      //  non-blocking stream -- it was seen zeroing rows of the first batch's level 0 after k_load_level0 had written them)
      if (hipMalloc((void**)&c->d_pyr_alt, (size_t)c->cfg.max_images * c->img_pitch) != hipSuccess ||
          hipMemsetAsync(c->d_pyr_alt, 0, (size_t)c->cfg.max_images * c->img_pitch, c->stream) != hipSuccess) {
        (void)hipGetLastError();
        c->d_pyr_alt = nullptr;
        c->pipeline_stereo = false;  // no room for the second pyramid: plain in-order execution
      }
    }
  }
  const bool piped = pipe && c->d_pyr_alt;
  if (piped)
    std::swap(c->d_pyr, c->d_pyr_alt);
  else
    TRY(join_stereo(c));
  {
    hipStream_t st = c->stream;
    // level 0 of slot 2p / 2p+1 <- left / right image p.  >= 32 images: the resize reads the caller's images itself and every workgroup
    // writes its block of level 0 into the pyramid from the tile it has staged anyway -- no copy-in kernel, half its traffic
    const bool ext0 = c->blur_stream && c->prof != 1 && !c->rs_regions.empty() && 2 * n_pairs >= 32 && (size_t)stride * c->cfg.height <= 0xFFFFFFF0u;
    ExtLevel0 ext;
    if (ext0) {
      ext.left = d_left, ext.right = d_right;
      ext.pitch = image_pitch, ext.stride = (int)stride, ext.bytes = (uint32_t)((size_t)stride * (c->cfg.height - 1) + c->cfg.width);
      ext.inputs_free = (pack && pack->in_free) ? pack->in_free : nullptr;
    } else {
      launch_load_level0(st, d_left, d_right, stride, image_pitch, c->d_pyr, c->img_pitch, L0.plane_off, L0.stride, c->cfg.width, c->cfg.height, 0, 2,
                         n_pairs);
      if (pack && pack->in_free) HIP_TRY(c, hipEventRecord(pack->in_free, st));  // the images may be overwritten
    }
    TRY(run_extract(c, st, 0, 2 * n_pairs, (piped && c->stereo_pending) ? c->ev_stereo_done : nullptr, true, ext0 ? &ext : nullptr));
    if (piped) {
      HIP_TRY(c, hipEventRecord(c->ev_brief_done, st));
      HIP_TRY(c, hipStreamWaitEvent(c->stereo_stream, c->ev_brief_done, 0));
      TRY(run_stereo(c, c->stereo_stream, 0, 1, 2, 0, n_pairs, fx, bf));
    } else {
      TRY(run_stereo(c, st, 0, 1, 2, 0, n_pairs, fx, bf));
    }
  }
  if (pack) {
    // the packed results of this batch -> the stream's result buffer (device to device: ~0.1 ms for 512 pairs), on the stream the
    // match ran on, BEFORE the next batch may rewrite the per-slot arrays; the download then runs beside the next batch
    hipStream_t ps = piped ? c->stereo_stream : c->stream;
    const PackLayout l = pack_layout(c, n_pairs);
    const size_t NF = (size_t)std::max(c->cfg.n_features, 1), n = (size_t)n_pairs;
    HIP_TRY(c, hipStreamWaitEvent(ps, pack->wait_free, 0));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_kps, c->d_kps, 2 * n * NF * sizeof(orbfe_keypoint), hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_desc, c->d_desc, 2 * n * NF * 32, hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_cnt, c->d_n_kp, 2 * n * 4, hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_ru, c->d_right_u, n * NF * 8, hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_dp, c->d_depth, n * NF * 8, hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_nm, c->d_n_match, n * 4, hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipEventRecord(pack->ready, ps));
  }
  if (piped) {
    HIP_TRY(c, hipEventRecord(c->ev_stereo_done, c->stereo_stream));
    c->stereo_pending = true;
  }
  return ORBFE_OK;
}

extern "C" {

orbfe_status orbfe_stereo_batch_device(orbfe_ctx* c, const uint8_t* d_left, const uint8_t* d_right, size_t stride, size_t image_pitch,
                                       int32_t n_pairs, float fx, float bf) {
  ApiLock api_lk(c);
  if (!c || !d_left || !d_right || n_pairs < 0) return fail(c, ORBFE_EBADARG, "stereo_batch_device: NULL argument");
  if (2 * n_pairs > c->cfg.max_images) return fail(c, ORBFE_ECAPACITY, "stereo_batch_device: %d pairs need %d slots > %d", n_pairs, 2 * n_pairs, c->cfg.max_images);
  if (stride < (size_t)c->cfg.width || image_pitch < stride * (size_t)c->cfg.height)
    return fail(c, ORBFE_EBADARG, "stereo_batch_device: stride/pitch too small");
  if (n_pairs == 0) return ORBFE_OK;
  TRY(slots_idle(c, 0, 2 * n_pairs, "stereo_batch_device"));
  HIP_TRY(c, hipSetDevice(c->device));
  return batch_device_core(c, d_left, d_right, stride, image_pitch, n_pairs, fx, bf, nullptr);
}

}  // extern "C"
