// orbfe_stream.hip -- page-locked host memory and the host-image stream: the frame loop of example/Stereo/KittiStereo.cc:28-37 over batches
// from host memory (orbfe_host_alloc*, orbfe_stream_submit / wait / device_results / pack_records).  (Split from orbfe_api.hip in r5.)
#include "orbfe_ctx.h"
extern "C" {

// ---- host-image stream ---------------------------------------------------------------------------------------------------------
// CPUs of the NUMA node the current HIP device hangs off (sysfs local_cpulist of its PCI function); empty set if unknown.
static bool device_local_cpus(int dev, cpu_set_t* set) {
  char bus[64] = {0};
  if ((dev < 0 && hipGetDevice(&dev) != hipSuccess) || hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  for (char* q = bus; *q; ++q) *q = (char)tolower(*q);
  char path[160];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
  FILE* f = fopen(path, "r");
  if (!f) return false;
  char line[1024] = {0};
  const bool got = fgets(line, sizeof line, f) != nullptr;
  fclose(f);
  if (!got) return false;
  CPU_ZERO(set);
  int n = 0;
  char* save = nullptr;  // (strtok_r: allocations may come from several threads at once)
  for (char* tok = strtok_r(line, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {
    int a = 0, b = 0;
    const int k = sscanf(tok, "%d-%d", &a, &b);
    if (k == 1) b = a;
    if (k < 1) continue;
    for (int c = a; c <= b && c < CPU_SETSIZE; ++c) {
      CPU_SET(c, set);
      ++n;
    }
  }
  return n > 0;
}

// Page-locked host memory ON THE NUMA NODE OF THE DEVICE: the pages are placed where the allocating thread runs, and a buffer on the
// other socket is read by the DMA engines across the inter-socket link (measured on a two-socket MI355X host: 41 GB/s instead of
// 57 GB/s host to device).  The calling thread is moved to the device's local CPUs for the allocation and the first touch, then back.
void* orbfe_host_alloc(size_t bytes) { return orbfe_host_alloc_on(-1, bytes); }

// device_id < 0: the calling thread's current HIP device; a multi-rank job passes its own device so that ranks that never called
// hipSetDevice do not all pin to GPU 0's node.
void* orbfe_host_alloc_on(int32_t device_id, size_t bytes) {
  cpu_set_t old_set, local;
  const bool have_old = sched_getaffinity(0, sizeof old_set, &old_set) == 0;
  bool moved = false;
  if (have_old && device_local_cpus(device_id, &local)) {
    cpu_set_t both;
    CPU_AND(&both, &local, &old_set);  // stay inside what this process is allowed to use
    if (CPU_COUNT(&both) > 0) moved = sched_setaffinity(0, sizeof both, &both) == 0;
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    p = nullptr;
  } else if (moved) {
    for (size_t o = 0; o < bytes; o += 4096) ((volatile uint8_t*)p)[o] = 0;  // first touch, should the driver place lazily
  }
  if (moved && sched_setaffinity(0, sizeof old_set, &old_set) != 0 && sched_setaffinity(0, sizeof old_set, &old_set) != 0)
    (void)fail(nullptr, ORBFE_OK, "orbfe_host_alloc: the calling thread's CPU affinity could not be restored (it stays on the device's NUMA node)");
  return p;
}
void orbfe_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

// The streaming entry points keep up to eight HIP streams busy at once (compute, stereo match, blur, upload, download, slot lanes, the
// caller's and RCCL's own), and the HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues -- 4 by
// default.  Two streams that share a queue run one after the other: with 4 queues the upload of batch k + 1 queues behind the kernels
// of batch k and the 4541-pair sequence takes 0.127 s, with 16 it takes 0.087 s (profiles/r3_hw_queues.txt).  The runtime reads the
// variable at its first HIP call, so a library cannot set it: the process that streams exports it (bench.py does; a process that builds
// one frame at a time should NOT -- 4 queues are ~50 us per frame faster there), and orbfe_stream_submit says so once if it is missing.
int32_t orbfe_recommended_hw_queues(void) { return 16; }

orbfe_status orbfe_stream_submit(orbfe_ctx* c, const uint8_t* left, const uint8_t* right, size_t stride, size_t image_pitch, int32_t n_pairs,
                                 float fx, float bf, const orbfe_batch_results* out, int64_t* ticket) {
  {
    static std::once_flag warned;
    std::call_once(warned, [] {
      const char* q = getenv("GPU_MAX_HW_QUEUES");
      if (!q || atoi(q) < 8)
        fprintf(stderr,
                "[orbfe] orbfe_stream_submit: GPU_MAX_HW_QUEUES is %s; the streaming path overlaps upload, compute and download on streams of their "
                "own and runs ~30 %% slower when they share hardware queues -- export GPU_MAX_HW_QUEUES=%d before the process's first HIP call "
                "(orbfe_recommended_hw_queues())\n",
                q ? q : "unset (4)", orbfe_recommended_hw_queues());
    });
  }
  ApiLock api_lk(c);
  if (!c || !left || !right || !out || !ticket || n_pairs <= 0) return fail(c, ORBFE_EBADARG, "stream_submit: NULL argument / no pairs");
  if (2 * n_pairs > c->cfg.max_images) return fail(c, ORBFE_ECAPACITY, "stream_submit: %d pairs need %d slots > %d", n_pairs, 2 * n_pairs, c->cfg.max_images);
  if (stride < (size_t)c->cfg.width || image_pitch < stride * (size_t)c->cfg.height)
    return fail(c, ORBFE_EBADARG, "stream_submit: stride/pitch too small");
  TRY(slots_idle(c, 0, 2 * n_pairs, "stream_submit"));
  HIP_TRY(c, hipSetDevice(c->device));
  orbfe_ctx::HostStream& hs = c->hs;
  if (!hs.init) {
    HIP_TRY(c, hipStreamCreateWithFlags(&hs.h2d, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&hs.d2h, hipStreamNonBlocking));
    for (int b = 0; b < orbfe_ctx::HostStream::kDepth; ++b) {
      HIP_TRY(c, hipEventCreateWithFlags(&hs.ev_h2d[b], hipEventDisableTiming));
      HIP_TRY(c, hipEventCreateWithFlags(&hs.ev_in_free[b], hipEventDisableTiming));
      HIP_TRY(c, hipEventCreateWithFlags(&hs.ev_out_ready[b], hipEventDisableTiming));
      HIP_TRY(c, hipEventCreateWithFlags(&hs.ev_done[b], hipEventDisableTiming));
    }
    hs.init = true;
  }
  const size_t eye = image_pitch * (size_t)n_pairs;
  const PackLayout l = pack_layout(c, n_pairs);
  if (2 * eye > hs.in_bytes || l.total > hs.out_bytes) {  // (re)size the device buffers: quiesce everything first
    TRY(join_stereo(c));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(hs.h2d));
    HIP_TRY(c, hipStreamSynchronize(hs.d2h));
    const size_t in_bytes = std::max(hs.in_bytes, align_up(2 * eye, 1 << 20));
    const PackLayout lmax = pack_layout(c, c->cfg.max_images / 2);
    const size_t out_bytes = std::max(hs.out_bytes, lmax.total);
    for (int b = 0; b < orbfe_ctx::HostStream::kDepth; ++b) {
      if (in_bytes != hs.in_bytes) {
        if (hs.d_in[b]) HIP_TRY(c, hipFree(hs.d_in[b]));
        hs.d_in[b] = nullptr;
        HIP_TRY(c, hipMalloc((void**)&hs.d_in[b], in_bytes));
      }
      if (out_bytes != hs.out_bytes) {
        if (hs.d_out[b]) HIP_TRY(c, hipFree(hs.d_out[b]));
        hs.d_out[b] = nullptr;
        HIP_TRY(c, hipMalloc((void**)&hs.d_out[b], out_bytes));
      }
    }
    hs.in_bytes = in_bytes;
    hs.out_bytes = out_bytes;
  }
  const int D = orbfe_ctx::HostStream::kDepth;
  const int b = (int)(hs.next_ticket % D);
  // the ticket that last used this set of buffers must be complete ("at most three outstanding" is what makes three sets enough)
  if (hs.next_ticket >= D && hipEventQuery(hs.ev_done[b]) != hipSuccess) {
    (void)hipGetLastError();
    HIP_TRY(c, hipEventSynchronize(hs.ev_done[b]));
  }
  // upload: after the batch that last read this input buffer has consumed it
  HIP_TRY(c, hipStreamWaitEvent(hs.h2d, hs.ev_in_free[b], 0));
  HIP_TRY(c, hipMemcpyAsync(hs.d_in[b], left, eye, hipMemcpyHostToDevice, hs.h2d));
  HIP_TRY(c, hipMemcpyAsync(hs.d_in[b] + eye, right, eye, hipMemcpyHostToDevice, hs.h2d));
  HIP_TRY(c, hipEventRecord(hs.ev_h2d[b], hs.h2d));
  // compute: the device-batch schedule, results packed into this ticket's result buffer
  HIP_TRY(c, hipStreamWaitEvent(c->stream, hs.ev_h2d[b], 0));
  const PackDst pack = {hs.d_out[b], hs.ev_done[b], hs.ev_out_ready[b], hs.ev_in_free[b]};
  TRY(batch_device_core(c, hs.d_in[b], hs.d_in[b] + eye, stride, image_pitch, n_pairs, fx, bf, &pack));
  // download
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1), n = (size_t)n_pairs;
  // (measured and dropped: writing the results into the page-locked arrays with a copy KERNEL through their device mapping instead of
  //  the DMA engine -- 16 to 1024 workgroups, four 16-byte loads in flight per lane: the step takes 11.7 ms against 9.1 ms.)
  // The download goes onto the stream the pack ran on.  In the pipelined schedule that is the stereo stream, which has nothing else
  // to do until the next batch's match ~8 ms later -- a stream of its own would be one more hardware queue, and HIP multiplexes all
  // streams of a process onto 4 of them (GPU_MAX_HW_QUEUES): a download that shares its queue with the uploads or with the compute
  // stream serialises with them (measured: 9.1 -> 11.8 ms per 512-pair step, depending on what the process had created before).
  hipStream_t ds = c->stereo_pending ? c->stereo_stream : hs.d2h;
  if (ds == hs.d2h) HIP_TRY(c, hipStreamWaitEvent(hs.d2h, hs.ev_out_ready[b], 0));
  const uint8_t* src = hs.d_out[b];
  if (out->kps) HIP_TRY(c, hipMemcpyAsync(out->kps, src + l.o_kps, 2 * n * NF * sizeof(orbfe_keypoint), hipMemcpyDeviceToHost, ds));
  if (out->desc) HIP_TRY(c, hipMemcpyAsync(out->desc, src + l.o_desc, 2 * n * NF * 32, hipMemcpyDeviceToHost, ds));
  if (out->counts) HIP_TRY(c, hipMemcpyAsync(out->counts, src + l.o_cnt, 2 * n * 4, hipMemcpyDeviceToHost, ds));
  if (out->right_u) HIP_TRY(c, hipMemcpyAsync(out->right_u, src + l.o_ru, n * NF * 8, hipMemcpyDeviceToHost, ds));
  if (out->depth) HIP_TRY(c, hipMemcpyAsync(out->depth, src + l.o_dp, n * NF * 8, hipMemcpyDeviceToHost, ds));
  if (out->n_matches) HIP_TRY(c, hipMemcpyAsync(out->n_matches, src + l.o_nm, n * 4, hipMemcpyDeviceToHost, ds));
  HIP_TRY(c, hipEventRecord(hs.ev_done[b], ds));
  hs.n_pairs_of[b] = n_pairs;
  *ticket = hs.next_ticket++;
  return ORBFE_OK;
}

orbfe_status orbfe_stream_wait(orbfe_ctx* c, int64_t ticket) {
  ApiLock api_lk(c);
  if (!c) return ORBFE_EBADARG;
  orbfe_ctx::HostStream& hs = c->hs;
  if (!hs.init || ticket < 0 || ticket >= hs.next_ticket) return fail(c, ORBFE_EBADARG, "stream_wait: ticket %lld was never issued", (long long)ticket);
  const int D = orbfe_ctx::HostStream::kDepth;
  if (ticket + D < hs.next_ticket) return ORBFE_OK;  // ticket + D has been submitted since, and that submit waited for this one
  HIP_TRY(c, hipSetDevice(c->device));
  hipEvent_t done = hs.ev_done[ticket % D];
  api_lk.lk.unlock();  // the wait itself needs nothing of the context: another thread may submit meanwhile
  HIP_TRY(c, hipEventSynchronize(done));
  return ORBFE_OK;
}

orbfe_status orbfe_stream_device_results(orbfe_ctx* c, int64_t ticket, int32_t n_pairs, const void** d_kps, const void** d_desc,
                                         const void** d_counts, const void** d_right_u, const void** d_depth, const void** d_nmatch) {
  ApiLock api_lk(c);
  if (!c) return ORBFE_EBADARG;
  orbfe_ctx::HostStream& hs = c->hs;
  if (!hs.init || ticket < 0 || ticket >= hs.next_ticket || ticket + orbfe_ctx::HostStream::kDepth < hs.next_ticket || n_pairs <= 0 ||
      2 * n_pairs > c->cfg.max_images)
    return fail(c, ORBFE_EBADARG, "stream_device_results: ticket %lld is not live (next %lld) or bad pair count %d", (long long)ticket,
                (long long)hs.next_ticket, n_pairs);
  if (n_pairs != hs.n_pairs_of[ticket % orbfe_ctx::HostStream::kDepth])
    return fail(c, ORBFE_EBADARG, "stream_device_results: ticket %lld was submitted with %d pairs, not %d (the packed layout depends on it)",
                (long long)ticket, hs.n_pairs_of[ticket % orbfe_ctx::HostStream::kDepth], n_pairs);
  const PackLayout l = pack_layout(c, n_pairs);
  const uint8_t* b = hs.d_out[ticket % orbfe_ctx::HostStream::kDepth];
  if (d_kps) *d_kps = b + l.o_kps;
  if (d_desc) *d_desc = b + l.o_desc;
  if (d_counts) *d_counts = b + l.o_cnt;
  if (d_right_u) *d_right_u = b + l.o_ru;
  if (d_depth) *d_depth = b + l.o_dp;
  if (d_nmatch) *d_nmatch = b + l.o_nm;
  return ORBFE_OK;
}

// Frame records of a ticket (layout: k_glue.hip, k_pack_records) into caller-provided DEVICE memory, for the sequence-level gather.
size_t orbfe_record_bytes(const orbfe_ctx* c) { return c ? 16 + (size_t)std::max(c->cfg.n_features, 1) * (28 + 32 + 8 + 8) : 0; }

orbfe_status orbfe_stream_pack_records(orbfe_ctx* c, int64_t ticket, int32_t n_pairs, void* d_records) {
  ApiLock api_lk(c);
  if (!c || !d_records) return fail(c, ORBFE_EBADARG, "stream_pack_records: NULL argument");
  orbfe_ctx::HostStream& hs = c->hs;
  if (!hs.init || ticket < 0 || ticket >= hs.next_ticket || ticket + orbfe_ctx::HostStream::kDepth < hs.next_ticket || n_pairs <= 0 ||
      2 * n_pairs > c->cfg.max_images)
    return fail(c, ORBFE_EBADARG, "stream_pack_records: ticket %lld is not live (next %lld) or bad pair count %d", (long long)ticket,
                (long long)hs.next_ticket, n_pairs);
  if (n_pairs != hs.n_pairs_of[ticket % orbfe_ctx::HostStream::kDepth])
    return fail(c, ORBFE_EBADARG, "stream_pack_records: ticket %lld was submitted with %d pairs, not %d (the packed layout depends on it)",
                (long long)ticket, hs.n_pairs_of[ticket % orbfe_ctx::HostStream::kDepth], n_pairs);
  HIP_TRY(c, hipSetDevice(c->device));
  const int b = (int)(ticket % orbfe_ctx::HostStream::kDepth);
  const PackLayout l = pack_layout(c, n_pairs);
  const uint8_t* src = hs.d_out[b];
  // on the download stream, behind the ticket's own completion: ordered after the results are in the buffer and before the
  // buffer is handed to ticket + 3 (whose pack waits for the event recorded here)
  HIP_TRY(c, hipStreamWaitEvent(hs.d2h, hs.ev_done[b], 0));
  launch_pack_records(hs.d2h, src + l.o_kps, src + l.o_desc, (const int32_t*)(src + l.o_cnt), src + l.o_ru, src + l.o_dp,
                      (const int32_t*)(src + l.o_nm), std::max(c->cfg.n_features, 1), n_pairs, d_records);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(hs.ev_done[b], hs.d2h));
  hipEvent_t done = hs.ev_done[b];
  api_lk.lk.unlock();  // the wait lasts a batch's compute and needs nothing of the context: another thread may submit / fetch meanwhile
  HIP_TRY(c, hipEventSynchronize(done));
  return ORBFE_OK;
}

}  // extern "C"
