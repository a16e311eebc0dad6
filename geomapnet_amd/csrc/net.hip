// MapNet / PoseNet training plan: the host-side schedule of one training step
// (/root/reference/common/train.py:322-363) as a fixed sequence of gfx950 kernel launches on
// caller-provided arenas.  See include/mapnet_hip.h for the ABI and DESIGN.md for the layout.
#include "../../include/mapnet_hip.h"

#include <cmath>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "criterion.h"
#include "elementwise.h"
#include "elementwise_h2.h"
#include "head.h"
#include "dense.h"
#include "igemm.h"
#include "dgrad.h"
#include "halo_pp.h"
#include "layout.h"
#include "optim.h"
#include "pool.h"
#include "stem.h"
#include "stem_bwd.h"
#include "util.h"
#include "wgrad.h"

using namespace mn;

// ---- model layout entry points ------------------------------------------------------------------
static const Layout& layout_for(int feat_dim) {
  static thread_local std::unique_ptr<Layout> cache;
  if (!cache || cache->feat_dim != feat_dim) cache.reset(new Layout(feat_dim));
  return *cache;
}
extern "C" int mn_model_entries(int feat_dim) { return (int)layout_for(feat_dim).entries.size(); }
extern "C" int mn_model_entry(int feat_dim, int idx, mn_entry* out) {
  const Layout& L = layout_for(feat_dim);
  if (idx < 0 || idx >= (int)L.entries.size()) return fail("mn_model_entry: index out of range");
  *out = L.entries[idx];
  return 0;
}
extern "C" int64_t mn_model_param_floats(int feat_dim) { return layout_for(feat_dim).param_floats; }
extern "C" int64_t mn_model_buffer_bytes(int feat_dim) { return layout_for(feat_dim).buffer_bytes; }

namespace {

struct Bump {
  size_t cur = 0;
  size_t take(size_t bytes) {
    size_t o = cur;
    cur += (bytes + 255) & ~(size_t)255;
    return o;
  }
};

// event-pair timer for kernel classes (igemm, wgrad, whole step)
struct KernelTimer {
  bool enabled = false;
  struct Pair {
    hipEvent_t a, b;
    int cat;
  };
  std::deque<Pair> pool;  // stable addresses: callers hold Pair* across later begin() calls
  size_t used = 0;
  float ms[4] = {0, 0, 0, 0};
  int launches[4] = {0, 0, 0, 0};
  Pair* begin(int cat, hipStream_t s) {
    if (!enabled) return nullptr;
    if (used == pool.size()) {
      Pair p;
      hipEventCreate(&p.a);
      hipEventCreate(&p.b);
      pool.push_back(p);
    }
    Pair* p = &pool[used++];
    p->cat = cat;
    hipEventRecord(p->a, s);
    return p;
  }
  void end(Pair* p, hipStream_t s) {
    if (p) hipEventRecord(p->b, s);
  }
  void reset() { used = 0; }
  void collect() {  // caller has synchronised the stream
    for (int i = 0; i < 4; ++i) {
      ms[i] = 0;
      launches[i] = 0;
    }
    for (size_t i = 0; i < used; ++i) {
      float t = 0;
      hipEventSynchronize(pool[i].b);
      hipEventElapsedTime(&t, pool[i].a, pool[i].b);
      ms[pool[i].cat] += t;
      launches[pool[i].cat]++;
    }
  }
  ~KernelTimer() {
    for (auto& p : pool) {
      hipEventDestroy(p.a);
      hipEventDestroy(p.b);
    }
  }
};

struct PlanBase {
  virtual ~PlanBase() {
    for (hipEvent_t e : fork_events) hipEventDestroy(e);
    if (overflow_host) hipHostFree(overflow_host);
  }
  // (one process-wide side stream per device, never destroyed: plans come and go, the stream is reused)
  static hipStream_t shared_side_stream() {
    static hipStream_t streams[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    // (a side stream of lower or higher priority was measured: no gain)
    // MN_WGRAD_CUS=n (experiment, round 5): the side stream may only use n of the chip's CUs (hipExtStreamCreateWithCUMask) -- a
    // static partition, so that weight-gradient workgroups cannot take every CU away from the main stream's BatchNorm /
    // data-gradient launches.  MN_WGRAD_CU_SPREAD=1 spreads the n CUs evenly over the mask's bit range instead of taking the first
    // n.  Measured: profiles/r05 (cu mask); not a default.
    if (!streams[dev]) {
      const int ncu = getenv("MN_WGRAD_CUS") ? atoi(getenv("MN_WGRAD_CUS")) : 0;
      int total = 0;
      hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev);
      hipError_t e;
      if (ncu > 0 && total > 0 && ncu < total) {
        const bool spread = getenv("MN_WGRAD_CU_SPREAD") && atoi(getenv("MN_WGRAD_CU_SPREAD")) != 0;
        std::vector<uint32_t> mask((total + 31) / 32, 0u);
        for (int i = 0; i < ncu; ++i) {
          const int bit = spread ? (int)((long)i * total / ncu) : i;
          mask[bit / 32] |= 1u << (bit % 32);
        }
        e = hipExtStreamCreateWithCUMask(&streams[dev], (uint32_t)mask.size(), mask.data());
      } else {
        e = hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking);
      }
      if (e != hipSuccess) {
        streams[dev] = nullptr;
        (void)hipGetLastError();
      }
    }
    return streams[dev];
  }
  // Weight-gradient launches do not feed the data-gradient chain, so they run on a second stream: their
  // workgroups fill the CUs the tail of each data-gradient launch leaves idle (layers 3-4 at B = 192 have
  // 1056 / 528 tiles for 512 resident slots).  fork: wstream waits for everything enqueued on s so far;
  // join: s waits for wstream.
  hipStream_t wstream = nullptr;
  std::vector<hipEvent_t> fork_events;
  size_t fork_next = 0;
  bool wgrad_pending = false;
  bool overlap_wgrad = !(getenv("MN_WGRAD_STREAM") && atoi(getenv("MN_WGRAD_STREAM")) == 0);
  hipEvent_t next_fork_event() {
    if (fork_events.size() < 64) {
      hipEvent_t e = nullptr;
      hipEventCreateWithFlags(&e, hipEventDisableTiming);
      fork_events.push_back(e);
      return e;
    }
    return fork_events[fork_next++ % fork_events.size()];
  }
  hipStream_t fork_wgrad(hipStream_t s) {
    if (!overlap_wgrad || s == nullptr || timer.enabled) return s;
    if (!wstream && !(wstream = shared_side_stream())) {
      overlap_wgrad = false;
      return s;
    }
    hipEvent_t e = next_fork_event();
    hipEventRecord(e, s);
    hipStreamWaitEvent(wstream, e, 0);
    wgrad_pending = true;
    if (head_wgrads_deferred) launch_head_wgrads(wstream);  // (head_backward: the first fork of the backward pass takes them along)
    return wstream;
  }
  bool head_wgrads_deferred = false;
  virtual void launch_head_wgrads(hipStream_t st) = 0;
  void join_wgrad(hipStream_t s) {
    if (!wgrad_pending) return;
    hipEvent_t e = next_fork_event();
    hipEventRecord(e, wstream);
    hipStreamWaitEvent(s, e, 0);
    wgrad_pending = false;
  }
  // Early loss read-back: the reference blocks on loss.item() (common/train.py:361) after the whole step; the value
  // exists as soon as the criterion has run, so it is copied to the caller's pinned host float right there and the
  // caller waits for THAT event only -- the host prepares the next step while backward and the optimiser still run.
  float* loss_host = nullptr;
  hipEvent_t loss_event = nullptr;
  bool loss_pending = false;
  void post_loss(const float* loss_dev_ptr, hipStream_t s) {
    if (!loss_host || s == nullptr) return;
    if (!loss_event && hipEventCreateWithFlags(&loss_event, hipEventDisableTiming) != hipSuccess) {
      loss_event = nullptr;
      (void)hipGetLastError();
      return;
    }
    hipMemcpyAsync(loss_host, loss_dev_ptr, sizeof(float), hipMemcpyDeviceToHost, s);
    hipEventRecord(loss_event, s);
    loss_pending = true;
  }
  // The step is enqueued DIRECTLY (~250 launches on two streams; the host needs ~1.5 ms per step and stays a full step
  // ahead of the device).  Capturing it into a hipGraph and replaying it was implemented and measured in rounds 1-2 and
  // removed in round 3: on ROCm 7.2 the replay was 2 % SLOWER than direct launches (19.57 vs 19.17 ms), forked branches
  // overlapped less inside a graph (early weight-gradient forks: +0.3 % in a graph, -2.4 % eager), and destroying a graph
  // exec that contains a fork/join crashed later graph launches in hip::Graph::UpdateStreams.
  virtual void after_optim_host() = 0;
  virtual int sync_step_to_device(hipStream_t s) = 0;
  virtual int64_t applied_steps() = 0;  // the device's count of optimiser steps that were applied (waits for the device)
  virtual int forward(const void* images, float* poses_out, int training, hipStream_t s) = 0;
  // dropout between the feature vector's ReLU and the pose heads (mn_set_dropout): probability, Philox key, calls so far
  float drop_p = 0.f;
  unsigned long long drop_seed = 0;
  unsigned drop_calls = 0;
  bool drop_this_step = false;  // the forward pass of the step being built applied a mask (the backward pass must, too)
  bool input_u8 = false;  // images are uint8 NHWC, normalised on the device (mn_set_input_u8)
  InputNorm input_norm{{1.f, 1.f, 1.f}, {0.f, 0.f, 0.f}};
  virtual int loss_only(const float* pred, const float* targ, float* loss_out, hipStream_t s) = 0;
  virtual int forward_loss(const void* images, const float* targets, float* loss_out, float* poses_out,
                           hipStream_t s) = 0;
  virtual int backward_stage(int stage, hipStream_t s) = 0;
  virtual int optim_step(float grad_mul, hipStream_t s) = 0;
  virtual int debug_tensor(const char* name, void** ptr, int64_t* numel, int32_t* dtype) = 0;
  virtual float* grad_arena() = 0;
  mn_config cfg;
  // fp16 loss scaling.  cur_scale multiplies d(pred) and is divided out where gradients enter the fp32 arena.  An
  // overflowed step is skipped on the device (optim.h, adam_prep_kernel); the count of skipped steps is copied to a
  // pinned host word after every optimiser step and read -- without waiting -- when the next step is enqueued: each
  // newly seen skip halves the scale, `scale_growth_interval` clean steps double it (0 = never grow).  The host reacts
  // one or two steps late; until then the device keeps skipping, so no non-finite value reaches weights or moments.
  float cur_scale = 1.f;
  bool overflow_guard = false;
  long long* overflow_host = nullptr;  // pinned: {last step skipped, skipped steps in total, last skipped attempt, last COMPLETED attempt}
  long long skipped_seen = 0;
  long long done_seen = 0;        // last completed attempt the host has accounted for (overflow_host[3])
  int64_t attempts = 0;           // optimiser steps enqueued on this plan (skipped ones included)
  int64_t scale_set_at = 0;       // `attempts` when the loss scale last changed: later attempts ran under cur_scale
  int64_t stuck_skips = 0;        // CONSECUTIVE skips seen while the scale already was 1 (nothing left to lower); an APPLIED
                                  // step, a scale growth or mn_set_loss_scale resets it
  long long clean_steps = 0;      // applied steps since the last skip the host has seen
  int scale_growth_interval = getenv("MN_SCALE_GROWTH") ? atoi(getenv("MN_SCALE_GROWTH")) : 2000;
  // The host reads the pinned words WITHOUT waiting for the device, so a poll that sees no new skip proves nothing by itself: the
  // host may simply be ahead (C-API callers that never read the loss, DataLoader jitter).  Progress is only what the device reports:
  // overflow_host[3], the index of the last attempt it completed.  (Round-4 ADVICE: the previous form reset `stuck_skips` and
  // advanced `clean_steps` on every poll without a new skip, so a run with permanently non-finite inputs whose polls interleave
  // with "nothing finished yet" could reset the counter forever and silently train nothing, and the loss scale grew on polls
  // that observed nothing.)
  void poll_overflow() {
    if (!overflow_guard || !overflow_host) return;
    const long long done = *(volatile long long*)(overflow_host + 3);   // read first: a copy landing in between can only make
    const long long seen = *(volatile long long*)(overflow_host + 1);   // `seen` newer than `done` (fewer clean steps counted)
    const long long last_bad = *(volatile long long*)(overflow_host + 2);
    const long long new_skips = seen > skipped_seen ? seen - skipped_seen : 0;
    const long long new_done = done > done_seen ? done - done_seen : 0;
    if (new_skips > 0) {
      // The host reads the count one or two steps late, and the steps enqueued meanwhile overflow under the OLD scale too:
      // a burst is ONE overflow event.  Halve once per burst -- i.e. only when a skipped step was enqueued after the
      // scale last changed.
      if (last_bad > scale_set_at) {
        if (cur_scale > 1.f) {
          cur_scale *= 0.5f;
          scale_set_at = attempts;
        } else {
          stuck_skips += new_skips;  // non-finite values that no loss scale can fix (inputs, forward pass)
        }
      }
      skipped_seen = seen;
      clean_steps = 0;
    }
    if (new_done > 0) {
      done_seen = done;
      // positive evidence of an applied step: the most recent completed attempt was not a skipped one
      if (last_bad < done) stuck_skips = 0;
      if (new_skips == 0) {
        clean_steps += new_done;
        if (scale_growth_interval > 0 && clean_steps >= scale_growth_interval) {
          clean_steps = 0;
          if (cur_scale < 65536.f) {
            cur_scale *= 2.f;
            scale_set_at = attempts;
          }
        }
      }
    }
  }
  Layout L{2048};
  float lr = 1e-4f, wd = 0.f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f, max_grad_norm = 0.f;
  int optim_method = 0, nesterov = 0;  // optim.h AdamArgs::method; beta1 / beta2 carry momentum / dampening | alpha
  int64_t step = 0;
  bool weights_dirty = true;
  int learn_beta = 0, learn_gamma = 0;
  KernelTimer timer;
};

template <typename T>
struct Plan : PlanBase {
  static constexpr int VEC = ElemTraits<T>::VEC;
  static constexpr int DT = ElemTraits<T>::DTYPE;

  struct Unit {  // conv + BatchNorm
    ConvP cp;
    BnP bp;
    int Hin, Win, Hout, Wout;
    long M;
    GatherGeom gf;
    DgradGeom dg;  // data-gradient launches (generic form + stride-2 parity classes), dgrad.h
    T* wf = nullptr;  // forward operand  [Cout][K]
    T* wd = nullptr;  // data-gradient operand [Cin][R*S*Cout]
    T* y = nullptr;   // raw conv output [M][Cout]
    T* gy = nullptr;  // its gradient
    half* rec = nullptr;  // fp16x2m: the BatchNorm's backward record [M][Cout] (elementwise_h2.h rec_pack), written by the forward apply
    float *mean, *invstd;
    float *coef_f, *coef_b;  // per-channel coefficients of the forward apply [2][C] / backward apply [4][C] (finalize kernels)
    double *accum_f, *accum_b;  // fp64 sums of the forward statistics / backward reductions, [rows_f | rows_b][2][C]
    int rows_f = 0, rows_b = 0; // accumulator rows (ACC_ROWS; MN_DETERMINISTIC: one row per producing workgroup)
    int ldw;           // row pitch of the master weight gradient
    const int* colmap = nullptr;
  };
  struct Block {
    Unit u1, u2, ud;
    bool down;
    int stage;
    const T* x;   // block input
    T* gx;        // gradient w.r.t. block input (written by this block's backward)
    T* a1;        // relu(bn1(conv1 x))
    T* ga1;
    T* zd;        // bn_d(conv_d x) when down
    T* out;       // block output
    T* gout;      // gradient w.r.t. block output (written by the consumer)
    // (fp16x2m, round 5: plain fp16 COPIES of the h2 activations fed the single-fp16 backward pass; round 6: its kernels read the hi
    //  halves of the h2 tensors themselves -- WgradArgs::x_h2, Epilogue::gate_h2 -- and the copies are gone: 2 bytes per activation
    //  element less written by every BatchNorm apply, 2.5 GB less arena at 192 images)
  };

  float *params, *grads, *m1, *m2;
  char* buffers;
  char* work;
  int B, H, W, Hp, Wp;
  int frames;  // images per window
  T* xpad;
  Unit stem;
  T *a0, *ga0, *p0, *gp0;
  half* xpad16 = nullptr;   // fp16x2m: fp16 copy of the padded NHWC4 input (the stem's fp16 backward kernels, stem_bwd.h)
  int H0, W0, H1, W1;  // stem conv output, pooled output
  std::vector<Block> blocks;
  int Hl, Wl;  // last feature map
  float *pooled, *feat, *poses, *dposes, *dz, *dpooled, *fcT, *loss_dev, *dropmask;
  // BatchNorm sums are accumulated with fp64 atomics straight from the producing kernels (conv epilogue, backward
  // reduction) into ACC_ROWS rows per unit (row = producer block % ACC_ROWS, to spread same-address contention);
  // the consuming apply kernels add the rows in their prologue.  No separate partial-reduction launches.
  // MN_DETERMINISTIC=1: bit-reproducible training steps -- every floating-point sum the step takes is added in an order
  // that does not depend on scheduling: BatchNorm sums through one accumulator row per producing workgroup (below),
  // weight gradients through per-split workspace slices and an ordered reduction (conv_wgrad), the gradient norm through
  // per-workgroup partial sums (optim.h).  Costs ~x ms per step.
  const bool deterministic = getenv("MN_DETERMINISTIC") && atoi(getenv("MN_DETERMINISTIC")) != 0;
  static constexpr int ACC_ROWS = 8;  // (32 rows measured equal: 14.73 vs 14.69 ms per step)
  double* acc_region = nullptr;
  size_t acc_bytes = 0;
  int cur_training = 1;
  double* sqnorm;             // squared gradient norm: the first double of acc_region
  bool sqnorm_clean = false;  // zeroed by this step's fill and not yet accumulated into
  static constexpr int kSqPartials = 8192;
  double* sq_partials = nullptr;  // MN_DETERMINISTIC: per-workgroup sums of the gradient norm
  unsigned char* frozen;
  int* stem_colmap;
  RepackJob* repack_jobs;
  int repack_njobs = 0, repack_blocks = 0, repack_head_jobs = -1, repack_head_blocks = 0;
  bool grads_zeroed = false;
  unsigned char* pool_idx;  // winning tap of every max-pool window
  float* wgf_ws = nullptr;  // partial tiles of the fused weight gradient (wgrad_fused.h), shared by its launches (one stream)
  long wgf_ws_floats = 0;
  void* zero_page;          // 256 zero bytes: source of out-of-image taps for the DMA conv pipeline
  long long* overflow_dev;  // {this step skipped, skipped steps in total, last skipped attempt, last completed attempt} (adam_prep_kernel)
  long long* step_dev;      // device-resident Adam step counter
  float* bc_dev;            // {1 - beta1^t, 1 - beta2^t}, derived on device from step_dev
  const float* cur_targets = nullptr;
  float* cur_loss = nullptr;

  // ---- construction -----------------------------------------------------------------------
  size_t carve(char* base) {
    Bump b;
    auto A = [&](size_t bytes) { return base ? base + b.take(bytes) : (b.take(bytes), (char*)nullptr); };
    // all BatchNorm accumulators live in one region so a single memset per step clears them
    auto set_rows = [&](Unit& u, bool is_stem) {
      u.rows_f = u.rows_b = ACC_ROWS;
      if (!deterministic) return;
      // one row per producing workgroup: every (row, channel) slot then receives exactly ONE atomic add onto zero, and the
      // finalize kernels add the rows in a fixed order.  Forward producers: the stem kernel's persistent workgroups, the
      // layer1 kernel's persistent workgroups (halo_pp.h: at most one per 16x16-pixel tile), M-tiles of >= 128 rows elsewhere.
      if (is_stem)
        u.rows_f = use_stem_kernel && (DT == MN_F16 || mma_fwd == MMA_F16X3) ? 1024 : cdiv((int)u.M, 128);
      else if (halo_path(u.gf))
        u.rows_f = u.gf.B * cdiv(u.gf.P, kHaloTH) * cdiv(u.gf.Q, kHaloTW);
      else
        u.rows_f = cdiv((int)u.M, 128);
      // (layer1's h2 convolutions on the persistent register-resident-weight kernel, halo_h2.h: one row per workgroup)
      if (h2 && !q8 && !is_stem && use_conv_halo_h2() && u.cp.k == 3 && u.cp.stride == 1 && u.cp.cin == 64 && u.cp.cout == 64) {
        const int wgs = conv_halo_h2_grid(u.gf);
        if (u.rows_f < wgs) u.rows_f = wgs;
      }
      u.rows_b = 1024;  // >= the workgroups of a backward reduction (launch_bn_bwd: ~512)
    };
    set_rows(stem, true);
    size_t acc_doubles = (size_t)(stem.rows_f + stem.rows_b) * 2 * 64;
    for (auto& blk : blocks) {
      set_rows(blk.u1, false);
      set_rows(blk.u2, false);
      if (blk.down) set_rows(blk.ud, false);
      acc_doubles += (size_t)(blk.u1.rows_f + blk.u1.rows_b + blk.u2.rows_f + blk.u2.rows_b) * 2 * blk.u1.cp.cout;
      if (blk.down) acc_doubles += (size_t)(blk.ud.rows_f + blk.ud.rows_b) * 2 * blk.ud.cp.cout;
    }
    // ... and the squared gradient norm rides at its head: ONE fill launch per step clears all of it (hipMemsetAsync's fill
    // kernel took ~110 us per call for these 2 MB, on the main stream at the head of every step; round-3 profile)
    acc_bytes = (acc_doubles + 2) * 8;
    acc_region = (double*)A(acc_bytes);
    sqnorm = acc_region;
    double* acc_cursor = acc_region ? acc_region + 2 : nullptr;
    auto unit_bufs = [&](Unit& u) {
      int C = u.cp.cout;
      u.y = (T*)A((size_t)u.M * C * sizeof(T));
      u.gy = (T*)A((size_t)u.M * C * sizeof(T));
      u.mean = (float*)A(C * 4);
      u.invstd = (float*)A(C * 4);
      u.coef_f = (float*)A(2 * C * 4);
      u.coef_b = (float*)A(4 * C * 4);
      u.accum_f = acc_cursor;
      u.accum_b = acc_cursor ? acc_cursor + (size_t)u.rows_f * 2 * C : nullptr;
      if (acc_cursor) acc_cursor += (size_t)(u.rows_f + u.rows_b) * 2 * C;
    };
    xpad = (T*)A((size_t)B * Hp * Wp * 4 * sizeof(T));
    // stem
    unit_bufs(stem);
    stem.wf = (T*)A((size_t)64 * 224 * sizeof(T));
    size_t n0 = (size_t)B * H0 * W0 * 64, n1 = (size_t)B * H1 * W1 * 64;
    a0 = (T*)A(n0 * sizeof(T));
    ga0 = (T*)A(n0 * sizeof(T));
    p0 = (T*)A(n1 * sizeof(T));
    gp0 = (T*)A(n1 * sizeof(T));
    pool_idx = (unsigned char*)A(n1);
    if (mixed) xpad16 = (half*)A((size_t)B * Hp * Wp * 4 * sizeof(half));
    const T* x = p0;
    T* gx = gp0;
    for (auto& blk : blocks) {
      blk.x = x;
      blk.gx = gx;
      Unit* us[3] = {&blk.u1, &blk.u2, blk.down ? &blk.ud : nullptr};
      for (Unit* u : us) {
        if (!u) continue;
        unit_bufs(*u);
        if (use_rec) u->rec = (half*)A((size_t)u->M * u->cp.cout * sizeof(half));
        size_t wn = (size_t)u->cp.cout * u->cp.cin * u->cp.k * u->cp.k;
        if (DT == MN_F16 || h2)
          u->wf = (T*)A(wn * sizeof(T));  // (h2: 4 bytes per element as well)
        else
          u->wf = base ? (T*)(params + u->cp.w) : nullptr;  // fp32: the OHWI master is the operand
        u->wd = (T*)A(wn * sizeof(T));
      }
      size_t no = (size_t)blk.u2.M * blk.u2.cp.cout;
      blk.a1 = (T*)A(no * sizeof(T));
      blk.ga1 = (T*)A(no * sizeof(T));
      blk.zd = blk.down ? (T*)A(no * sizeof(T)) : nullptr;
      blk.out = (T*)A(no * sizeof(T));
      blk.gout = (T*)A(no * sizeof(T));
      x = blk.out;
      gx = blk.gout;
    }
    int F = cfg.feat_dim;
    pooled = (float*)A((size_t)B * 512 * 4);
    feat = (float*)A((size_t)B * F * 4);
    poses = (float*)A((size_t)B * 6 * 4);
    dposes = (float*)A((size_t)B * 6 * 4);
    dz = (float*)A((size_t)B * F * 4);
    dropmask = (float*)A((size_t)B * F * 4);
    dpooled = (float*)A((size_t)B * 512 * 4);
    fcT = (float*)A((size_t)512 * F * 4);
    loss_dev = (float*)A(256);
    sq_partials = (double*)A((size_t)kSqPartials * 8);
    frozen = (unsigned char*)A(256);
    stem_colmap = (int*)A(224 * 4);
    repack_jobs = (RepackJob*)A(64 * sizeof(RepackJob));
    zero_page = (void*)A(256);
    wgf_ws_floats = (DT == MN_F16 || mma_bwd == MMA_BF16X3) ? wgrad_fused_ws_floats(WGF_BLOCKS) : 0;  // (all fused forms)
    if (deterministic && wgf_ws_floats < (16L << 20)) wgf_ws_floats = 16L << 20;  // split slices of the plain weight gradients
    wgf_ws = wgf_ws_floats ? (float*)A((size_t)wgf_ws_floats * 4) : nullptr;
    step_dev = (long long*)A(256);
    bc_dev = (float*)A(256);
    overflow_dev = (long long*)A(256);
    return b.cur;
  }

  static GatherGeom fwd_geom(int B, int Hin, int Win, const ConvP& c, int Hout, int Wout) {
    GatherGeom g;
    g.B = B; g.Hi = Hin; g.Wi = Win; g.C = c.cin; g.P = Hout; g.Q = Wout; g.R = c.k; g.S = c.k;
    g.mul_p = c.stride; g.mul_q = c.stride; g.rsign = 1; g.ssign = 1; g.off_h = -c.pad; g.off_w = -c.pad; g.div = 1;
    g.M = B * Hout * Wout; g.N = c.cout; g.K = c.k * c.k * c.cin;
    return g;
  }
  void init_unit(Unit& u, const ConvP& c, const BnP& b, int Hin, int Win) {
    u.cp = c;
    u.bp = b;
    u.Hin = Hin;
    u.Win = Win;
    u.Hout = (Hin + 2 * c.pad - c.k) / c.stride + 1;
    u.Wout = (Win + 2 * c.pad - c.k) / c.stride + 1;
    u.M = (long)B * u.Hout * u.Wout;
    u.gf = fwd_geom(B, Hin, Win, c, u.Hout, u.Wout);
    u.ldw = c.k * c.k * c.cin;
    u.dg = make_dgrad_geom(B, Hin, Win, c.cin, c.cout, c.k, c.stride, c.pad, u.Hout, u.Wout, VEC);
    u.gf.mma = mma_fwd;
    u.dg.full.mma = mma_bwd;
    for (int i = 0; i < u.dg.n_pc; ++i) u.dg.pc[i].g.mma = mma_bwd;
  }

  // MN_DTYPE_F32X3 (Plan<float> only): fp32 tensors, contractions on the f16 / bf16 matrix pipe with split operands
  // (common.h MMA_*): forward f16x3, backward bf16x3; the fc / pose head stay on the exact fp32 MFMA
  int mma_fwd = MMA_NATIVE, mma_bwd = MMA_NATIVE;
  // MN_DTYPE_F16X2 (Plan<float> only): every tensor a convolution consumes -- block inputs / outputs, a1, the projection
  // output, d(conv output), both weight copies -- is an h2 tensor (fp16 hi + lo halves, 4 bytes per element: the arena is
  // carved exactly as for fp32), written by the h2 element-wise kernels (elementwise_h2.h) and read by the h2 convolution
  // kernels through LDS-DMA; raw conv outputs, data gradients, statistics, head, criterion and optimiser are fp32.  The stem
  // (3 input channels: no 32-channel groups) stays on the fp32x3 kernels: fp32 xpad / stem.gy, f16x3 / bf16x3 contraction.
  // Gradients live in fp16 pairs, so the mode scales the loss and guards against overflow exactly as the fp16 mode does.
  bool h2 = false;
  // MN_DTYPE_F16X2M (round 5; Plan<float> only): the fp16x2 FORWARD pass, bit for bit -- loss and poses are that mode's -- and a
  // backward pass that contracts SINGLE fp16 operands, one MFMA per product, on the fp16 mode's kernels: d(conv output), data
  // gradients and the data-gradient weight copy are plain fp16 tensors (in the buffers the arena carves for fp32), the X operand of
  // a weight gradient and the ReLU gates read plain fp16 COPIES of the activations that the h2 producers write beside the h2 tensor.
  // What stays exact: every gate (sign of the forward value) and BatchNorm's backward statistics (fp32 conv output, fp64 sums).
  // What it costs the gradients: 1.1e-3 relative L2 overall / 1.6e-3 worst tensor (tools/mixed_budget.py: the oracle's own step
  // with fp16 conv operands in the backward pass) -- a fifth of the 4.9e-3 / 1.1e-2 by which two correct fp32 evaluations of this
  // network differ through ReLU gate flips.  The STEM is the exception to "statistics from the exact forward
  // values": by default (stem_f16) its two backward kernels (stem_bwd.h) read an fp16 copy of the fp32 conv output (written by the
  // stem's BatchNorm + max-pool pass) and an fp16 image of the input, so its BatchNorm-backward sums and xhat come from fp16-rounded
  // y; only its ReLU gate stays exact (the pooled gradient arrives already gated by layer1.0's data gradient).  MN_DETERMINISTIC=1
  // or MN_STEM_BWD=0 keep the fp32 chain of fp16x2 for the stem.
  bool mixed = false;
  // fp16x2m, round 6: BatchNorm's backward pass reads a 2-byte RECORD of the forward pass (fp16 xhat + the ReLU's outcome in its
  // lowest bit, written by the forward apply) instead of the 4-byte conv output -- 2 bytes per element more in one forward pass, 2
  // fewer in each of the two backward passes, and the gate stays exact.  MN_BN_REC=0: round 5's form (fp32 conv output re-read, gate
  // recomputed), for same-box A/Bs.
  bool use_rec = false;
  // MN_DTYPE_F16X2Q (round 5; Plan<float> only): fp16x2m whose FORWARD convolutions take both cross terms of a split-operand product
  // from fp8 copies on the block-scaled MFMA (common.h MMA_H2Q, h2q tensors): 2 instead of 3 MFMA-equivalents per forward product,
  // poses ~3e-4 from the fp32 reference instead of 1.6e-5 (bar 1e-3)
  bool q8 = false;
  std::string stage_error;  // set by a launch helper of a backward stage that cannot return a status itself
  Plan(const mn_config& c) {
    cfg = c;
    cur_scale = c.loss_scale;
    q8 = DT == MN_F32 && c.dtype == MN_DTYPE_F16X2Q;
    mixed = DT == MN_F32 && (c.dtype == MN_DTYPE_F16X2M || q8);
    h2 = DT == MN_F32 && (c.dtype == MN_DTYPE_F16X2 || mixed);
    stem_f16 = mixed && !deterministic && !(getenv("MN_STEM_BWD") && atoi(getenv("MN_STEM_BWD")) == 0);
    use_rec = mixed && !(getenv("MN_BN_REC") && atoi(getenv("MN_BN_REC")) == 0);
    if (c.dtype == MN_DTYPE_F32X3 || h2) {  // (fp16x2m: the stem's backward and its fp32 tensors)
      mma_fwd = MMA_F16X3;
      mma_bwd = MMA_BF16X3;
    }
    overflow_guard = (DT == MN_F16 || h2) && !(getenv("MN_OVERFLOW_GUARD") && atoi(getenv("MN_OVERFLOW_GUARD")) == 0);
    // (fp16x2m: 0 / 1 / 2 measured equal in round 5; with round 6's shorter BatchNorm-backward passes 1 leads by 0.3 %: 18.96 / 19.02 /
    //  19.15 ms for 1 / 0 / 2, two interleaved repeats, profiles/r06/c6_*)
    // (fp16: one fork per block since round 4; round 6, with the per-stage choices below: 13.10 -> 13.00 ms, four of four interleaved
    //  pairs, profiles/r06/c52_to_c54_*)
    if (!getenv("MN_WGRAD_SCHED") && early_fork) wgrad_sched = (mixed || h2 || DT == MN_F16) ? 1 : 2;  // (see wgrad_sched)
    const bool staged_choice = mixed || (DT == MN_F16 && !h2);
    if (!getenv("MN_WGRAD_EARLY_STAGES") && staged_choice) wgrad_early_stages = 13;  // (see wgrad_early_stages)
    if (!getenv("MN_WGRAD_DEFER_STAGES") && staged_choice) wgrad_defer_stages = 2;   // (see wgrad_defer_stages)
    L = Layout(c.feat_dim);
    frames = (c.mode == MN_MODE_POSENET) ? 1 : (c.mode == MN_MODE_MAPNET ? c.T : 2 * c.T);
    B = c.windows * frames;
    H = c.H;
    W = c.W;
    Hp = H + 6;
    Wp = (W + 6 + 1 + 1) & ~1;  // even, >= W + 7
    // stem as a 7x4 conv over pixel pairs (C = 8), stride (2,1), no padding
    H0 = (H + 6 - 7) / 2 + 1;
    W0 = (W + 6 - 7) / 2 + 1;
    stem.cp = L.stem;
    stem.bp = L.stem_bn;
    stem.Hin = H; stem.Win = W; stem.Hout = H0; stem.Wout = W0;
    stem.M = (long)B * H0 * W0;
    GatherGeom g;
    g.B = B; g.Hi = Hp; g.Wi = Wp / 2; g.C = 8; g.P = H0; g.Q = W0; g.R = 7; g.S = 4;
    g.mul_p = 2; g.mul_q = 1; g.rsign = 1; g.ssign = 1; g.off_h = 0; g.off_w = 0; g.div = 1;
    g.M = B * H0 * W0; g.N = 64; g.K = 224;
    g.mma = mma_fwd;
    stem.gf = g;
    stem.ldw = 147;
    H1 = (H0 + 2 - 3) / 2 + 1;
    W1 = (W0 + 2 - 3) / 2 + 1;
    int h = H1, w = W1;
    for (const BlockP& bp : L.blocks) {
      Block blk;
      blk.down = bp.down;
      blk.stage = bp.stage;
      init_unit(blk.u1, bp.c1, bp.b1, h, w);
      init_unit(blk.u2, bp.c2, bp.b2, blk.u1.Hout, blk.u1.Wout);
      if (bp.down) init_unit(blk.ud, bp.cd, bp.bd, h, w);
      h = blk.u1.Hout;
      w = blk.u1.Wout;
      blocks.push_back(blk);
    }
    Hl = h;
    Wl = w;
  }

  int attach(float* params_, float* opt_state, void* buffers_, void* work_, hipStream_t s) {
    params = params_;
    grads = opt_state;
    m1 = opt_state + L.param_floats;
    m2 = opt_state + 2 * L.param_floats;
    buffers = (char*)buffers_;
    work = (char*)work_;
    size_t total = carve(work);
    hipMemsetAsync(work, 0, total, s);
    // stem column map: compute column (r, s4, e) -> dense OHWI column (r, s', ch) or -1
    std::vector<int> cm(224);
    for (int r = 0; r < 7; ++r)
      for (int s4 = 0; s4 < 4; ++s4)
        for (int e = 0; e < 8; ++e) {
          int sp = 2 * s4 + (e >> 2), ch = e & 3;
          cm[(r * 4 + s4) * 8 + e] = (sp < 7 && ch < 3) ? (r * 7 + sp) * 3 + ch : -1;
        }
    hipMemcpyAsync(stem_colmap, cm.data(), 224 * 4, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);  // cm is a host temporary
    stem.colmap = stem_colmap;
    if (overflow_guard && !overflow_host) {
      if (hipHostMalloc((void**)&overflow_host, 4 * sizeof(long long)) != hipSuccess) {
        overflow_host = nullptr;
        (void)hipGetLastError();
      } else {
        overflow_host[0] = overflow_host[1] = overflow_host[2] = overflow_host[3] = 0;
      }
    }
    build_repack_table(s);
    update_frozen(s);
    weights_dirty = true;
    return check_launch("attach");
  }

  void update_frozen(hipStream_t s) {
    unsigned char f[4] = {(unsigned char)!learn_beta, (unsigned char)!learn_beta, (unsigned char)!learn_gamma,
                          (unsigned char)!learn_gamma};
    hipMemcpyAsync(frozen, f, 4, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
  }

  // ---- weights: fp32 master -> compute layouts ------------------------------------------------
  void build_repack_table(hipStream_t s) {
    std::vector<RepackJob> jobs;
    int blk = 0;
    auto add = [&](long src, void* a, void* b2, int O, int R, int S, int I, int mode) {
      RepackJob j;
      j.src_off = src; j.dst_a = a; j.dst_b = b2; j.O = O; j.R = R; j.S = S; j.I = I; j.mode = mode;
      long total = mode == 2 ? (long)O * R * 32 : (long)O * R * S * I;
      j.blk0 = blk;
      j.nblk = (int)((total + 4095) / 4096);
      blk += j.nblk;
      jobs.push_back(j);
    };
    add(stem.cp.w, stem.wf, nullptr, 64, 7, 7, 3, 2);
    repack_head_jobs = -1;
    for (auto& bk : blocks) {
      if (bk.stage >= 1 && repack_head_jobs < 0) {  // jobs before this one serve the stem and layer1
        repack_head_jobs = (int)jobs.size();
        repack_head_blocks = blk;
      }
      Unit* us[3] = {&bk.u1, &bk.u2, bk.down ? &bk.ud : nullptr};
      for (Unit* u : us) {
        if (!u) continue;
        const ConvP& c = u->cp;
        add(c.w, (DT == MN_F16 || h2) ? (void*)u->wf : nullptr, u->wd, c.cout, c.k, c.k, c.cin, 0);
      }
    }
    add(L.fc_w, fcT, nullptr, cfg.feat_dim, 1, 1, 512, 3);
    repack_njobs = (int)jobs.size();
    repack_blocks = blk;
    hipMemcpyAsync(repack_jobs, jobs.data(), jobs.size() * sizeof(RepackJob), hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);  // `jobs` is a host temporary
  }
  // head = stem + layer1 weights (needed first), tail = everything else; the tail can run on the side stream
  // while the stem and layer1 execute
  void repack_head(hipStream_t s) {
    const int hj = repack_head_jobs > 0 ? repack_head_jobs : repack_njobs;
    const int hb = repack_head_jobs > 0 ? repack_head_blocks : repack_blocks;
    if constexpr (DT == MN_F32) {
      if (q8) {  // forward operand h2q, data-gradient operand plain fp16
        hipLaunchKernelGGL((repack_all_kernel<float, true, true, true>), dim3(hb), dim3(256), 0, s, (const RepackJob*)repack_jobs, hj,
                           (const float*)params, 0);
        return;
      }
      if (mixed) {  // forward operand h2, data-gradient operand plain fp16
        hipLaunchKernelGGL((repack_all_kernel<float, true, true>), dim3(hb), dim3(256), 0, s, (const RepackJob*)repack_jobs, hj,
                           (const float*)params, 0);
        return;
      }
      if (h2) {
        hipLaunchKernelGGL((repack_all_kernel<float, true>), dim3(hb), dim3(256), 0, s, (const RepackJob*)repack_jobs, hj,
                           (const float*)params, 0);
        return;
      }
    }
    hipLaunchKernelGGL((repack_all_kernel<T>), dim3(hb), dim3(256), 0, s, (const RepackJob*)repack_jobs, hj,
                       (const float*)params, 0);
  }
  void repack_tail(hipStream_t s) {
    const int hb = repack_head_jobs > 0 ? repack_head_blocks : repack_blocks;
    if (repack_blocks > hb) {
      bool done = false;
      if constexpr (DT == MN_F32) {
        if (q8) {
          hipLaunchKernelGGL((repack_all_kernel<float, true, true, true>), dim3(repack_blocks - hb), dim3(256), 0, s,
                             (const RepackJob*)repack_jobs, repack_njobs, (const float*)params, hb);
          done = true;
        } else if (mixed) {
          hipLaunchKernelGGL((repack_all_kernel<float, true, true>), dim3(repack_blocks - hb), dim3(256), 0, s,
                             (const RepackJob*)repack_jobs, repack_njobs, (const float*)params, hb);
          done = true;
        } else if (h2) {
          hipLaunchKernelGGL((repack_all_kernel<float, true>), dim3(repack_blocks - hb), dim3(256), 0, s,
                             (const RepackJob*)repack_jobs, repack_njobs, (const float*)params, hb);
          done = true;
        }
      }
      if (!done)
        hipLaunchKernelGGL((repack_all_kernel<T>), dim3(repack_blocks - hb), dim3(256), 0, s,
                           (const RepackJob*)repack_jobs, repack_njobs, (const float*)params, hb);
    }
    weights_dirty = false;
  }

  // ---- forward ------------------------------------------------------------------------------------
  BnParams bn_params(const Unit& u) {
    BnParams p;
    p.gamma = params + u.bp.gamma;
    p.beta = params + u.bp.beta;
    p.running_mean = (float*)buffers + u.bp.rm;
    p.running_var = (float*)buffers + u.bp.rv;
    p.num_batches_tracked = (long long*)(buffers + u.bp.nbt);
    p.mean = u.mean;
    p.invstd = u.invstd;
    p.eps = 1e-5f;
    p.momentum = 0.1f;
    return p;
  }
  void conv_bn_stats(Unit& u, const T* x, int training, hipStream_t s) {
    Epilogue ep;
    ep.out = u.y; ep.ldc = u.cp.cout; ep.stats = nullptr; ep.bias = nullptr; ep.relu = 0;
    ep.res = nullptr; ep.res_gate = nullptr; ep.alpha = 1.f;
    if (training) {
      ep.stats_accum = u.accum_f;
      ep.stats_rows = u.rows_f;
    }
    auto* tp = timer.begin(0, s);
    if (h2 && &u != &stem)  // h2 activation and weights in, fp32 conv output + statistics out
      launch_igemm_h2(u.gf, (const half*)x, (const half*)u.wf, ep, s, (const half*)zero_page, q8);
    else if (&u == &stem && DT == MN_F32 && mma_fwd == MMA_F16X3 && use_stem_kernel)  // the split-operand form of the stem kernel
      launch_stem_conv_x3((const float*)x, (const float*)u.wf, (float*)u.y, training ? u.accum_f : nullptr, u.rows_f, B, H, W, Wp, s);
    else if (&u == &stem && DT == MN_F16 && use_stem_kernel)  // weights in registers, input pairs read straight from LDS (stem.h)
      launch_stem_conv((const half*)x, (const half*)u.wf, (half*)u.y, training ? u.accum_f : nullptr, u.rows_f, B, H, W, Wp, s);
    else if (halo_path(u.gf) && conv_halo_pp_applies(u.gf, ep))
      launch_conv_halo_pp(u.gf, (const half*)x, (const half*)u.wf, ep, s);
    else
      launch_igemm<T>(u.gf, x, u.wf, ep, s, (const T*)zero_page);
    timer.end(tp, s);
  }
  // layer1's 64-channel 3x3 convolutions (forward and data gradient) run from an LDS-resident input halo in the
  // persistent two-group kernel of halo_pp.h (MN_HALO=0: the implicit-GEMM kernel, for A/B measurements)
  bool use_halo = DT == MN_F16 && !(getenv("MN_HALO") && atoi(getenv("MN_HALO")) == 0);
  bool use_halo_bwd = !(getenv("MN_HALO") && atoi(getenv("MN_HALO")) == 0);  // fp16x2m: layer1's data gradients
  bool use_stem_kernel = !(getenv("MN_STEM_KERNEL") && atoi(getenv("MN_STEM_KERNEL")) == 0);
  // MN_DENSE=0: the head's fc layer through igemm.h's 128 x 128 tiles (parity tests, A/B)
  bool use_dense = !(getenv("MN_DENSE") && atoi(getenv("MN_DENSE")) == 0);
  bool halo_path(const GatherGeom& g) const { return use_halo && conv_halo_applies(g); }
  void bn_finalize(Unit& u, hipStream_t s) {  // statistics -> (scale, shift), mean / invstd, running statistics
#ifdef MN_ABLATION_BUILD
    static const int skip_c = getenv("MN_ABL_SKIP_FINALIZE") ? atoi(getenv("MN_ABL_SKIP_FINALIZE")) : 0;  // (see launch_bn_bwd)
    static long calls = 0;
    if (u.cp.cout <= skip_c && ++calls > 400) return;
#endif
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(u.cp.cout, kBnFinalizeChannels)), dim3(256), 0, s, (const double*)u.accum_f, (double)u.M,
                       bn_params(u), cur_training, u.coef_f, u.cp.cout, u.rows_f);
  }
  void bn_act(Unit& u, const T* res, int relu, T* out, hipStream_t s) {
    long np = u.M * u.cp.cout / VEC;
    bn_finalize(u, s);
    if (h2) {  // fp32 conv output in, h2 activation out (residual: an h2 activation)
      const long ni = u.M * u.cp.cout / 8;
      hipLaunchKernelGGL(bn_apply_h2_kernel, dim3(ew_grid(ni)), dim3(256), 0, s, (const float*)u.y, (const float*)u.coef_f,
                         (const half*)res, (half*)out, ni, u.cp.cout, relu, q8 ? 1 : 0,
                         cur_training ? u.rec : (half*)nullptr, (const float*)u.mean, (const float*)u.invstd);
      return;
    }
    hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(ew_grid(np)), dim3(256), 0, s, (const T*)u.y, (const float*)u.coef_f, res, out,
                       np, u.cp.cout, relu);
  }

  int forward(const void* images, float* poses_out, int training, hipStream_t s) override {
    return forward_impl(images, poses_out, training, false, s);
  }
  int forward_impl(const void* images, float* poses_out, int training, bool zero_grads, hipStream_t s) {
    // work the stem and layer1 do not depend on goes to the side stream: the repack of the later layers'
    // weights and optim.learner.zero_grad(); joined before layer2
    const bool dirty = weights_dirty;
    if (dirty) repack_head(s);
    grads_zeroed = zero_grads;
    cur_training = training;
    if (training) {  // forward statistics + backward reduction sums + the squared gradient norm
      launch_zero_fill(reinterpret_cast<float*>(acc_region), (long)(acc_bytes / 4), s);
      sqnorm_clean = true;
    }
    // (fp16x2m: the stem's fp16 backward kernels read an fp16 image of the input -- written by the same launch)
    half* const x16 = stem_bwd_f16() && training ? xpad16 : (half*)nullptr;
    if (input_u8)
      hipLaunchKernelGGL((u8nhwc_to_padded_nhwc4_kernel<T>), dim3(ew_grid((long)B * Hp * Wp)), dim3(256), 0, s,
                         (const unsigned char*)images, xpad, B, H, W, Hp, Wp, input_norm, x16);
    else
      hipLaunchKernelGGL((nchw_to_padded_nhwc4_kernel<T>), dim3(ew_grid((long)B * Hp * Wp)), dim3(256), 0, s,
                         (const float*)images, xpad, B, H, W, Hp, Wp, x16);
    // forked AFTER the (HBM-bound) input conversion, so that the side stream's copies run beside the stem's MFMA-bound
    // convolution instead of competing with the conversion for bandwidth (step time: equal within noise)
    if (dirty || zero_grads) {
      hipStream_t side = fork_wgrad(s);
      if (dirty) repack_tail(side);
      if (zero_grads) launch_zero_fill(grads, L.param_floats, side);
    }
    conv_bn_stats(stem, xpad, training, s);
    if (h2) {  // (always the fused form: fp32 conv output in, h2 pooled activation out)
      bn_finalize(stem, s);
      hipLaunchKernelGGL(bn_relu_maxpool_h2_kernel, dim3(ew_grid((long)B * H1 * W1 * 64 / 8)), dim3(256), 0, s,
                         (const float*)stem.y, (const float*)stem.coef_f, (half*)p0, pool_idx, B, H0, W0, 64, H1, W1,
                         training && stem_bwd_f16() ? (half*)a0 : (half*)nullptr, q8 ? 1 : 0);
    } else if (fuse_stem) {  // BatchNorm + ReLU + max-pool in one pass; the normalised stem activation is never stored
      bn_finalize(stem, s);
      hipLaunchKernelGGL((bn_relu_maxpool_kernel<T>), dim3(ew_grid((long)B * H1 * W1 * 64 / VEC)), dim3(256), 0, s,
                         (const T*)stem.y, (const float*)stem.coef_f, p0, pool_idx, B, H0, W0, 64, H1, W1);
    } else {
      bn_act(stem, nullptr, 1, a0, s);
      hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(ew_grid((long)B * H1 * W1 * 64 / VEC)), dim3(256), 0, s,
                         (const T*)a0, p0, pool_idx, B, H0, W0, 64, H1, W1);
    }
    for (auto& blk : blocks) {
      if (blk.stage >= 1) join_wgrad(s);  // no-op once joined
      // The projection path of a down block (1x1 stride-2 convolution, BatchNorm) depends on the block's input only: it runs on the
      // side stream -- idle in the forward pass -- beside conv1 -> bn1 -> conv2 and is joined before the residual add.  Three blocks,
      // ~0.25 ms of small launches: fp16x2m 18.18 -> 18.05 ms, four of four interleaved pairs (profiles/r06/c41_c42_*).
      // MN_FWD_DS_SIDE=0: in line, as before (A/B).
      static const bool ds_side = !(getenv("MN_FWD_DS_SIDE") && atoi(getenv("MN_FWD_DS_SIDE")) == 0);
      const bool side = blk.down && ds_side;
      if (side) {
        hipStream_t ws = fork_wgrad(s);
        conv_bn_stats(blk.ud, blk.x, training, ws);
        bn_act(blk.ud, nullptr, 0, blk.zd, ws);
      }
      conv_bn_stats(blk.u1, blk.x, training, s);
      bn_act(blk.u1, nullptr, 1, blk.a1, s);
      conv_bn_stats(blk.u2, blk.a1, training, s);
      const T* res = blk.x;
      if (blk.down) {
        if (side) {
          join_wgrad(s);
        } else {
          conv_bn_stats(blk.ud, blk.x, training, s);
          bn_act(blk.ud, nullptr, 0, blk.zd, s);
        }
        res = blk.zd;
      }
      bn_act(blk.u2, res, 1, blk.out, s);
    }
    join_wgrad(s);
    const Block& last = blocks.back();
    int F = cfg.feat_dim;
    if (h2)
      hipLaunchKernelGGL(avgpool_fwd_h2_kernel, dim3(cdiv((long)B * 512, 256)), dim3(256), 0, s, (const half*)last.out, pooled,
                         B, Hl * Wl, 512, q8 ? 1 : 0);
    else
      hipLaunchKernelGGL((avgpool_fwd_kernel<T>), dim3(cdiv((long)B * 512, 256)), dim3(256), 0, s, (const T*)last.out, pooled,
                         B, Hl * Wl, 512);
    // fc 512 -> feat_dim, bias, ReLU (models/posenet.py:46,65-66); dropout is the identity under the
    // reference's pinned torch 0.4.1 (F.dropout default training=False)
    GatherGeom g;
    g.B = B; g.Hi = 1; g.Wi = 1; g.C = 512; g.P = 1; g.Q = 1; g.R = 1; g.S = 1; g.mul_p = 1; g.mul_q = 1; g.rsign = 1;
    g.ssign = 1; g.off_h = 0; g.off_w = 0; g.div = 1; g.M = B; g.N = F; g.K = 512;
    Epilogue ep;
    ep.out = feat; ep.ldc = F; ep.stats = nullptr; ep.bias = params + L.fc_b; ep.relu = 1; ep.res = nullptr;
    ep.res_gate = nullptr; ep.alpha = 1.f;
    DenseArgs da;
    da.A = pooled; da.W = params + L.fc_w; da.bias = params + L.fc_b; da.C = feat; da.M = B; da.N = F; da.K = 512; da.lda = 512;
    da.ldw = 512; da.ldc = F; da.relu = 1;
    if (use_dense && dense_nt_applies(da))  // 32 x 32 tiles, K split over the workgroup's waves (dense.h)
      launch_dense_nt(da, s);
    else
      launch_igemm<float>(g, (const float*)pooled, (const float*)(params + L.fc_w), ep, s, (const float*)zero_page);
    // F.dropout(x, p=droprate) (models/posenet.py:68-69), in training mode only (mn_set_dropout explains the reference's two
    // readings): a fresh Philox mask per training forward pass, kept for the backward pass
    drop_this_step = training && drop_p > 0.f;
    if (drop_this_step) {
      const long n = (long)B * F;
      hipLaunchKernelGGL(dropout_fwd_kernel, dim3(cdiv((n + 3) / 4, 256)), dim3(256), 0, s, feat, dropmask, n, drop_p,
                         (unsigned)(drop_seed & 0xffffffffu), (unsigned)(drop_seed >> 32), drop_calls);
      drop_calls += 1;
    }
    hipLaunchKernelGGL(head_fwd_kernel, dim3(cdiv((long)B * 6 * 64, 256)), dim3(256), 0, s, (const float*)feat,
                       (const float*)(params + L.xyz_w), (const float*)(params + L.xyz_b),
                       (const float*)(params + L.wpqr_w), (const float*)(params + L.wpqr_b), poses, B, F);
    if (poses_out) hipMemcpyAsync(poses_out, poses, (size_t)B * 6 * 4, hipMemcpyDeviceToDevice, s);
    return check_launch("forward");
  }

  void run_criterion(const float* pred, const float* targ, float* loss, float* dpred, float* ds, hipStream_t s) {
    CriterionArgs a;
    a.mode = cfg.mode; a.N = cfg.windows; a.T = cfg.T; a.pred = pred; a.targ = targ; a.s = params + L.crit; a.loss = loss;
    a.dpred = dpred; a.ds = ds; a.vos_out = nullptr; a.grad_scale = cur_scale;
    hipLaunchKernelGGL(criterion_kernel, dim3(1), dim3(256), 0, s, a);
  }
  int loss_only(const float* pred, const float* targ, float* loss_out, hipStream_t s) override {
    run_criterion(pred, targ, loss_out, nullptr, nullptr, s);
    return check_launch("loss");
  }
  int forward_loss(const void* images, const float* targets, float* loss_out, float* poses_out, hipStream_t s) override {
    if (int e = forward_impl(images, poses_out, 1, true, s)) return e;
    cur_targets = targets;
    cur_loss = loss_out ? loss_out : loss_dev;
    return 0;
  }

  // ---- backward -------------------------------------------------------------------------------------
  // self_gate: `gate` is relu(bn_u(y)) itself (a1 of a block, a0 of the stem): recomputed from y, not read
  void bn_bwd(Unit& u, const T* g, const T* gate, hipStream_t s, bool self_gate = false) {
    if (h2 && &u != &stem && gate && !self_gate) {
      // (the h2 / fp16x2m apply kernels have no gate input -- block-output gradients arrive already gated -- so a
      //  caller that passes a real gate must not get ungated gradients silently)
      stage_error = "bn_bwd: the fp16x2 / fp16x2m modes take block-output gradients as stored (already gated); only self_gate exists";
      return;  // (reported by backward_stage: this call site has no status to return)
    }
    if (mixed && &u != &stem && u.rec) {  // fp16 gradient and the forward pass's record in; plain fp16 d(conv output) out
      launch_bn_bwd_rec((const half*)g, (const half*)u.rec, u.M, u.cp.cout, params + u.bp.gamma, u.mean, u.invstd, grads + u.bp.gamma,
                        grads + u.bp.beta, (half*)u.gy, u.accum_b, u.coef_b, 1.f / cur_scale, s, self_gate, u.rows_b);
      return;
    }
    if (mixed && &u != &stem) {  // (MN_BN_REC=0) fp16 gradient, fp32 conv output in; plain fp16 d(conv output) out
      launch_bn_bwd<half, float>((const half*)g, (const half*)nullptr, (const float*)u.y, u.M, u.cp.cout, params + u.bp.gamma, u.mean,
                                 u.invstd, grads + u.bp.gamma, grads + u.bp.beta, (half*)u.gy, u.accum_b, u.coef_b, 1.f / cur_scale, s,
                                 self_gate ? params + u.bp.beta : nullptr, PoolGradSrc(), u.rows_b);
      return;
    }
    if (h2 && &u != &stem) {  // fp32 gradient and conv output in, h2 d(conv output) out; gates: none, or the unit's own ReLU
      launch_bn_bwd_h2((const float*)g, (const float*)u.y, u.M, u.cp.cout, params + u.bp.gamma, u.mean, u.invstd,
                       grads + u.bp.gamma, grads + u.bp.beta, (half*)u.gy, u.accum_b, u.coef_b, 1.f / cur_scale, s,
                       self_gate ? params + u.bp.beta : nullptr, u.rows_b);
      return;
    }
    launch_bn_bwd<T>(g, gate, (const T*)u.y, u.M, u.cp.cout, params + u.bp.gamma, u.mean, u.invstd, grads + u.bp.gamma,
                     grads + u.bp.beta, u.gy, u.accum_b, u.coef_b, 1.f / cur_scale, s,
                     self_gate ? params + u.bp.beta : nullptr, PoolGradSrc(), u.rows_b);
  }
  // bit 0: BatchNorm+ReLU+max-pool in one forward pass (-0.15 ms/step); bit 1: max-pool gradient gathered inside the
  // BatchNorm backward passes instead of a maxpool_bwd launch (measured +0.05 ms/step: the gather runs twice) -- off
  int fuse_stem_mask = getenv("MN_FUSE_STEM") ? atoi(getenv("MN_FUSE_STEM")) : 1;
  bool fuse_stem = (fuse_stem_mask & 1) != 0;
  bool fuse_stem_bwd = (fuse_stem_mask & 2) != 0;
  bool early_fork = !(getenv("MN_EARLY_FORK") && atoi(getenv("MN_EARLY_FORK")) == 0);
  static constexpr bool parity_dgrad = true;  // stride-2 data gradients by parity class (dgrad.h: -1.6 % step time)
  // `ws`: the stream the launch goes to (the side stream after a fork, or the main stream)
  void conv_wgrad(Unit& u, const T* x, hipStream_t ws) {
    WgradArgs a;
    a.g = u.gf; a.dY = u.gy; a.ldy = u.cp.cout; a.X = x; a.dW = grads + u.cp.w; a.ldw = u.ldw; a.colmap = u.colmap;
    a.g.mma = mma_bwd;
    a.alpha = 1.f / cur_scale; a.rows_per_split = 0;
    a.ws = wgf_ws; a.ws_floats = wgf_ws_floats;  // every weight-gradient launch of a step goes to the same stream (`ws`)
    a.det = deterministic;
    auto* tp = timer.begin(1, ws);
    // reduction splits (measured, tools/conv_bench.py): one round of 2 workgroups per CU for the wide layers (half
    // the atomic traffic of 1024), more for layer1 and the stem whose pixel dimension is 4-16x longer
    const int target = u.cp.cout >= 128 ? 512 : (u.M > 2000000 ? 2048 : 1024);
    if (mixed && &u != &stem) {  // plain fp16 d(conv output) against the HI halves of the h2 activation: the fp16 mode's kernels
      a.g.mma = MMA_NATIVE;
      a.x_h2 = true;
      launch_wgrad<half>(a, target, ws, zero_page);
    } else if (h2 && &u != &stem) {  // h2 d(conv output) and activation
      a.g.mma = MMA_H2;
      launch_wgrad<half>(a, target, ws, zero_page);
    } else
    launch_wgrad<T>(a, target, ws, zero_page);
    timer.end(tp, ws);
  }
  void conv_dgrad(Unit& u, T* gx, const T* res, const T* gate, hipStream_t s, const T* out_gate = nullptr) {
    Epilogue ep;
    ep.out = gx; ep.ldc = u.cp.cin; ep.stats = nullptr; ep.bias = nullptr; ep.relu = 0; ep.res = res; ep.res_gate = gate;
    ep.out_gate = out_gate;
    ep.alpha = 1.f;
    auto* tp = timer.begin(0, s);
    if (mixed) {  // the fp16 mode's launches: fp16 d(conv output), weights, gradient and residual; the gate = hi halves of the h2 activation
      ep.gate_h2 = true;
      if (use_halo_bwd && conv_halo_applies(u.dg.full) && conv_halo_pp_applies(u.dg.full, ep))
        launch_conv_halo_pp(u.dg.full, (const half*)u.gy, (const half*)u.wd, ep, s);
      else
        launch_conv_dgrad<half>(u.dg, (const half*)u.gy, (const half*)u.wd, ep, s, (const half*)zero_page, parity_dgrad);
    } else if (h2)  // h2 d(conv output) and weights in, fp32 gradient out (+ fp32 residual, h2 gate of the block below)
      launch_conv_dgrad<half>(u.dg, (const half*)u.gy, (const half*)u.wd, ep, s, (const half*)zero_page, parity_dgrad, true);
    else if (halo_path(u.dg.full) && conv_halo_pp_applies(u.dg.full, ep))
      launch_conv_halo_pp(u.dg.full, (const half*)u.gy, (const half*)u.wd, ep, s);
    else
      launch_conv_dgrad<T>(u.dg, (const T*)u.gy, (const T*)u.wd, ep, s, (const T*)zero_page, parity_dgrad);
    timer.end(tp, s);
  }
  // Weight-gradient schedule (MN_WGRAD_SCHED): 0 = one fork per block, after its last BatchNorm backward;
  // 1 = each weight gradient forked as soon as its dY exists (starts beside the data gradient that consumes the
  // same dY); 2 = deferred: a weight gradient is queued and forked right before the NEXT BatchNorm-backward pass of
  // the main stream, so that the MFMA-bound launch starts beside HBM-bound work instead of beside a data gradient.
  // Round 2 (fused weight gradient: one 512-thread, 96 KB workgroup per CU -- it does not share a CU with a data-gradient
  // workgroup, the two time-slice): 1 = 16.44 ms, 0 = 16.05, 2 = 16.01 (same-box A/B); round 1's plain-GEMM weight gradient
  // preferred 1.  Round 4, after the fused weight gradient moved to its low-register forms (134 / 150 registers, 64 KB of LDS:
  // BatchNorm waves and, in part, other workgroups now fit beside it) the order turned over, same-box A/Bs, arms interleaved twice
  // (profiles/r04/c43_*): fp16 2 = 13.14 ms, 1 = 13.01, 0 = 12.98; fp16x2 2 = 28.60, 1 = 28.02, 0 = 28.38.  Defaults since:
  // fp16 0, fp16x2 1 (set in the constructor), fp32 / fp32x3 2 (not re-measured).
  int wgrad_sched = getenv("MN_WGRAD_SCHED") ? atoi(getenv("MN_WGRAD_SCHED")) : (early_fork ? 2 : 0);
  // (block_backward.  Same-box A/Bs, arms interleaved, profiles/r06/c29_to_c32_*: fp16x2m 0 -> 1: -0.04 ... -0.10 ms in six of six pairs
  //  (18.37 -> 18.31), 2: equal, 3: +0.06; fp16 12.78 -> 12.73.  3 with the stem's weight gradient at ONE workgroup per CU
  //  (MN_STEM_WGRAD_PER_CU=1, which alone costs +0.35 ms) is within 0.03 ms of 1.)
  int wgrad_tail = getenv("MN_WGRAD_TAIL") ? atoi(getenv("MN_WGRAD_TAIL")) : 1;
  // MN_WGRAD_EARLY_STAGES: bit k = schedule 1's early fork applies to stage k (else one fork per block, schedule 0's order).  fp16x2m
  // (set in the constructor): 13 = every stage but layer2, whose data gradients run in the 70 KB two-workgroup form and share their
  // CUs with whatever the side stream brings -- 18.61 -> 18.55 ms, six of six interleaved pairs (profiles/r06/c35_to_c37_*).
  int wgrad_early_stages = getenv("MN_WGRAD_EARLY_STAGES") ? atoi(getenv("MN_WGRAD_EARLY_STAGES")) : 15;
  // MN_WGRAD_DEFER_STAGES: bit k = stage k takes schedule 2's order (a weight gradient queued and forked right before the NEXT
  // BatchNorm-backward pass) whatever the plan's schedule is.  fp16x2m (set in the constructor): 2 = layer2 -- against one fork per
  // block there: 18.25 -> 18.20 ms, six of six interleaved pairs; layer4 +0.17, layer3 +0.05, layer1 +0.05 (profiles/r06/c38_to_c40_*).
  int wgrad_defer_stages = getenv("MN_WGRAD_DEFER_STAGES") ? atoi(getenv("MN_WGRAD_DEFER_STAGES")) : 0;
  struct PendingWgrad {
    Unit* u;
    const T* x;
  };
  std::vector<PendingWgrad> pending_wgrads;
  void flush_wgrads(hipStream_t s) {
    if (pending_wgrads.empty()) return;
    hipStream_t ws = fork_wgrad(s);
    for (auto& p : pending_wgrads) conv_wgrad(*p.u, p.x, ws);
    pending_wgrads.clear();
  }
  // Every gradient w.r.t. a block OUTPUT is stored already multiplied by that block's ReLU gate (out > 0): its producer
  // -- the data gradient of the block above, or the average pool's backward for the last block -- applies the gate in
  // its epilogue (Epilogue::out_gate, reading its own input tensor), so bn2 / the projection's BatchNorm / the identity
  // path of this block use `gout` as it is and never read `out`.
  void block_backward(Block& blk, hipStream_t s) {
    // the gate of the block below = ReLU that produced this block's input (none below layer1.0: its input is the max-pool)
    // (fp16x2m: gates and the weight gradients' X operands are the HI halves of the h2 activations, read in place by the fp16
    //  kernels: conv_wgrad sets WgradArgs::x_h2, conv_dgrad Epilogue::gate_h2)
    const T* bx = blk.x;
    const T* ba1 = blk.a1;
    // (fp16x2m with the fp16 stem kernels: the gradient of the pooled stem activation leaves layer1.0 gated by that activation)
    const T* below = &blk == &blocks.front() ? (stem_bwd_f16() ? bx : nullptr) : bx;
    const T* og = nullptr;  // (bn2 / the projection / the identity path take `gout` as stored: already gated)
    if (wgrad_sched == 2 || ((wgrad_defer_stages >> blk.stage) & 1)) {
      flush_wgrads(s);
      bn_bwd(blk.u2, blk.gout, og, s);
      if (!stage_error.empty()) return;
      conv_dgrad(blk.u2, blk.ga1, nullptr, nullptr, s);
      pending_wgrads.push_back({&blk.u2, ba1});
      flush_wgrads(s);
      bn_bwd(blk.u1, blk.ga1, ba1, s, true);
      if (blk.down) bn_bwd(blk.ud, blk.gout, og, s);
      pending_wgrads.push_back({&blk.u1, bx});
      if (blk.down) {
        pending_wgrads.push_back({&blk.ud, bx});
        conv_dgrad(blk.u1, blk.gx, nullptr, nullptr, s, below);  // gated here too: the projection launch below may only touch the even pixels
        conv_dgrad(blk.ud, blk.gx, blk.gx, nullptr, s, below);
      } else {
        conv_dgrad(blk.u1, blk.gx, blk.gout, og, s, below);
      }
      return;
    }
    bn_bwd(blk.u2, blk.gout, og, s);
    if (!stage_error.empty()) return;  // (round-5 ADVICE: do not run data / weight gradients on a d(conv output) that was never written)
    // MN_WGRAD_TAIL = k: the weight gradients of the first k blocks of layer1 (the LAST k of the backward pass) are held back and
    // forked in front of the stem's backward kernels (backward_stage: flush_wgrads), which run alone on the step stream at the end of
    // the step and are VALU-bound -- the matrix pipe is idle under them.  Same stage, same gradient bucket.
    const bool tail = blk.stage == 0 && !blk.down && (int)(&blk - &blocks[0]) < wgrad_tail;
    const bool early = wgrad_sched == 1 && !tail && ((wgrad_early_stages >> blk.stage) & 1);
    if (early) conv_wgrad(blk.u2, ba1, fork_wgrad(s));
    conv_dgrad(blk.u2, blk.ga1, nullptr, nullptr, s);
    bn_bwd(blk.u1, blk.ga1, ba1, s, true);
    if (blk.down) bn_bwd(blk.ud, blk.gout, og, s);
    if (tail) {
      pending_wgrads.push_back({&blk.u2, ba1});
      pending_wgrads.push_back({&blk.u1, bx});
      conv_dgrad(blk.u1, blk.gx, blk.gout, og, s, below);
      return;
    }
    hipStream_t ws = fork_wgrad(s);
    if (!early) conv_wgrad(blk.u2, ba1, ws);
    conv_wgrad(blk.u1, bx, ws);
    if (blk.down) {
      conv_wgrad(blk.ud, bx, ws);
      conv_dgrad(blk.u1, blk.gx, nullptr, nullptr, s, below);  // gated here too: the projection launch below may only touch the even pixels
      conv_dgrad(blk.ud, blk.gx, blk.gx, nullptr, s, below);  // accumulate the projection path in place, then gate
    } else {
      conv_dgrad(blk.u1, blk.gx, blk.gout, og, s, below);  // + identity path (already gated)
    }
  }
  void launch_head_wgrads(hipStream_t st) override {
    const int F = cfg.feat_dim;
    const float unscale = 1.f / cur_scale;
    hipLaunchKernelGGL(head_bwd_weight_kernel, dim3(cdiv(F + 1, 64)), dim3(256), 0, st, (const float*)dposes,
                       (const float*)feat, grads + L.xyz_w, grads + L.xyz_b, grads + L.wpqr_w, grads + L.wpqr_b, B, F,
                       unscale, cfg.filter_nans);
    // fc backward: weight + bias gradient in one launch of 32 x 32 tiles (dense.h)
    DenseWgradArgs dw;
    dw.dY = dz; dw.X = pooled; dw.dW = grads + L.fc_w; dw.db = grads + L.fc_b; dw.B = B; dw.F = F; dw.Cin = 512; dw.ldy = F;
    dw.ldx = 512; dw.ldw = 512; dw.alpha = unscale;
    launch_dense_wgrad(dw, st);
    head_wgrads_deferred = false;
  }
  void head_backward(hipStream_t s) {
    int F = cfg.feat_dim;
    float unscale = 1.f / cur_scale;
    if (!grads_zeroed) launch_zero_fill(grads, L.param_floats, s);  // optim.learner.zero_grad()
    grads_zeroed = false;
    run_criterion(poses, cur_targets, cur_loss, dposes, grads + L.crit, s);
    post_loss(cur_loss, s);
    hipLaunchKernelGGL(head_bwd_input_kernel, dim3(cdiv((long)B * F, 256)), dim3(256), 0, s, (const float*)dposes,
                       (const float*)feat, (const float*)(params + L.xyz_w), (const float*)(params + L.wpqr_w), dz, B, F,
                       cfg.filter_nans, drop_this_step ? (const float*)dropmask : (const float*)nullptr);
    // The head's and the fc layer's WEIGHT gradients feed nothing downstream: with two streams they leave with the first fork of the
    // backward pass (fork_wgrad), off the chain criterion -> heads -> fc -> average pool -> layer4 (MN_HEAD_WGRAD_SIDE=0: in line)
    static const bool head_side = !(getenv("MN_HEAD_WGRAD_SIDE") && atoi(getenv("MN_HEAD_WGRAD_SIDE")) == 0);
    if (head_side && overlap_wgrad && s != nullptr && !timer.enabled)
      head_wgrads_deferred = true;
    else
      launch_head_wgrads(s);
    GatherGeom g;
    g.B = B; g.Hi = 1; g.Wi = 1; g.C = 512; g.P = 1; g.Q = 1; g.R = 1; g.S = 1; g.mul_p = 1; g.mul_q = 1; g.rsign = 1;
    g.ssign = 1; g.off_h = 0; g.off_w = 0; g.div = 1; g.M = B; g.N = F; g.K = 512;
    GatherGeom gd = g;
    gd.C = F; gd.N = 512; gd.K = F;
    Epilogue ep;
    ep.out = dpooled; ep.ldc = 512; ep.stats = nullptr; ep.bias = nullptr; ep.relu = 0; ep.res = nullptr;
    ep.res_gate = nullptr; ep.alpha = 1.f;
    DenseArgs dd;
    dd.A = dz; dd.W = fcT; dd.bias = nullptr; dd.C = dpooled; dd.M = B; dd.N = 512; dd.K = F; dd.lda = F; dd.ldw = F; dd.ldc = 512;
    dd.relu = 0;
    if (use_dense && dense_nt_applies(dd))
      launch_dense_nt(dd, s);
    else
      launch_igemm<float>(gd, (const float*)dz, (const float*)fcT, ep, s, (const float*)zero_page);
    Block& last = blocks.back();
    if (mixed)  // fp16 gradient out, gate = the hi halves of the last block's h2 output
      hipLaunchKernelGGL((avgpool_bwd_kernel<half>), dim3(ew_grid((long)B * Hl * Wl * 512)), dim3(256), 0, s,
                         (const float*)dpooled, (half*)last.gout, B, Hl * Wl, 512, (const half*)last.out, 1);
    else if (h2)
      hipLaunchKernelGGL(avgpool_bwd_h2_kernel, dim3(ew_grid((long)B * Hl * Wl * 512)), dim3(256), 0, s, (const float*)dpooled,
                         (float*)last.gout, B, Hl * Wl, 512, (const half*)last.out);
    else
      hipLaunchKernelGGL((avgpool_bwd_kernel<T>), dim3(ew_grid((long)B * Hl * Wl * 512)), dim3(256), 0, s,
                         (const float*)dpooled, last.gout, B, Hl * Wl, 512, (const T*)last.out, 0);
  }
  // (MN_DETERMINISTIC: the stem's backward goes through bn_bwd + the split-slice weight gradient instead)
  bool use_stem_bwd = DT == MN_F16 && !deterministic && !(getenv("MN_STEM_BWD") && atoi(getenv("MN_STEM_BWD")) == 0);
  // fp16x2m: the same two kernels on fp16 COPIES of the fp32 conv output (written by the stem's BatchNorm + max-pool pass) and of
  // the padded input, with the pooled gradient arriving already gated (StemBwdArgs::pre_gated); MN_STEM_BWD=0 / MN_DETERMINISTIC
  // keep the fp32 chain of fp16x2
  bool stem_f16 = false;  // (set in the constructor, once `mixed` is known)
  bool stem_bwd_f16() const { return stem_f16; }
  void stem_backward(hipStream_t s) {
    if (use_stem_bwd || stem_bwd_f16()) {
      const bool mx = stem_bwd_f16();
      // two tile-walking launches (stem_bwd.h): BatchNorm sums, then the weight gradient with d(conv output) computed tile
      // by tile in LDS -- neither the max-pool's input gradient nor d(conv output) is stored
      StemBwdArgs a;
      a.y = mx ? (const half*)a0 : (const half*)stem.y; a.idx = pool_idx; a.gp = (const half*)gp0; a.gamma = params + stem.bp.gamma;
      a.beta = params + stem.bp.beta; a.coef = stem.coef_b; a.mean = stem.mean; a.invstd = stem.invstd; a.accum = stem.accum_b;
      a.accum_rows = stem.rows_b; a.xpad = mx ? (const half*)xpad16 : (const half*)xpad; a.dW = grads + stem.cp.w;
      a.colmap = stem_colmap; a.ldw = stem.ldw;
      a.alpha = 1.f / cur_scale;
      a.pre_gated = mx ? 1 : 0;
      launch_stem_bn_reduce(a, B, H, W, Wp, s);
      hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(64 / kBnFinalizeChannels), dim3(256), 0, s, (const double*)stem.accum_b, (double)stem.M,
                         (const float*)(params + stem.bp.gamma), (const float*)stem.mean, (const float*)stem.invstd,
                         grads + stem.bp.gamma, grads + stem.bp.beta, 1.f / cur_scale, (const float*)(params + stem.bp.beta),
                         stem.coef_b, 64, stem.rows_b);
      auto* tp = timer.begin(1, s);
      launch_stem_wgrad(a, B, H, W, Wp, s);
      timer.end(tp, s);
      return;
    }
    if (fuse_stem_bwd || h2) {  // (fp16x2: the stem's fp32 tensors make the materialised 1 GB gradient cost 0.5 ms per step)
      // the max-pool's input gradient is gathered from (argmax, pooled gradient) inside the BatchNorm backward
      // passes and the ReLU gate is recomputed from y: neither the activation nor its gradient exists in memory
      PoolGradSrc pg;
      pg.idx = pool_idx; pg.gout = gp0; pg.H = H0; pg.W = W0; pg.Po = H1; pg.Qo = W1;
      if (mixed) {  // layer1.0's data gradient left gp0 in fp16: widened into ga0 (unused by the fused stem) for the fp32 chain
        const long n1 = (long)B * H1 * W1 * 64;
        hipLaunchKernelGGL(widen_f16_kernel, dim3(ew_grid(n1 / 8)), dim3(256), 0, s, (const half*)gp0, (float*)ga0, n1 / 8);
        pg.gout = ga0;
      }
      launch_bn_bwd<T>((const T*)nullptr, (const T*)nullptr, (const T*)stem.y, stem.M, 64, params + stem.bp.gamma, stem.mean,
                       stem.invstd, grads + stem.bp.gamma, grads + stem.bp.beta, stem.gy, stem.accum_b, stem.coef_b,
                       1.f / cur_scale, s, params + stem.bp.beta, pg, stem.rows_b);
    } else {
      hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(ew_grid((long)B * H0 * W0 * 64 / VEC)), dim3(256), 0, s,
                         (const unsigned char*)pool_idx, (const T*)gp0, ga0, B, H0, W0, 64, H1, W1);
      bn_bwd(stem, ga0, a0, s, true);
    }
    // The input gradient of the stem is not needed (nothing consumes it).  The weight gradient goes to the stream every
    // other weight gradient of the step runs on: they share the partial-tile / split-slice workspace (wgf_ws), and the
    // last layer1 launches forked a moment ago may still be using it (joined at the end of the stage).
    conv_wgrad(stem, xpad, fork_wgrad(s));
  }
  int backward_stage(int stage, hipStream_t s) override {
    if (stage < 0 || stage > 3) return fail("backward_stage: stage must be 0..3");
    if (!cur_targets) return fail("backward_stage: call mn_train_forward_loss first");
    if (stage == 3) head_backward(s);
    // (a launch helper that refuses its arguments -- bn_bwd -- records stage_error: nothing that depends on its output is launched)
    for (int i = (int)blocks.size() - 1; i >= 0 && stage_error.empty(); --i)
      if (blocks[i].stage == stage) block_backward(blocks[i], s);
    if (stage == 0 && stage_error.empty()) {
      flush_wgrads(s);  // beside the stem's BatchNorm backward
      stem_backward(s);
    }
    flush_wgrads(s);
    if (head_wgrads_deferred) launch_head_wgrads(s);  // (no fork took them along)
    join_wgrad(s);  // the stage's gradient bucket is complete when this call's work on s is
    if (!stage_error.empty()) {  // a launch helper refused its arguments (bn_bwd): the stage is incomplete
      const std::string e = stage_error;
      stage_error.clear();
      return fail(e);
    }
    return check_launch("backward_stage");
  }

  float* grad_arena() override { return grads; }
  // ---- inspection (mn_debug_tensor): activations / gradients of the last step by name ---------------------------
  int debug_tensor(const char* name, void** ptr, int64_t* numel, int32_t* dtype) override {
    const std::string n(name);
    auto give = [&](const void* p, long count, int dt) {
      *ptr = const_cast<void*>(p);
      *numel = count;
      *dtype = dt;
      return 0;
    };
    const long n0 = (long)B * H0 * W0 * 64, n1 = (long)B * H1 * W1 * 64;
    if (n == "xpad") return give(xpad, (long)B * Hp * Wp * 4, DT);
    if (n == "stem.y") return give(stem.y, n0, DT);
    if (n == "stem.gy") return give(stem.gy, n0, DT);
    const int AT = q8 ? MN_DTYPE_F16X2Q : h2 ? MN_DTYPE_F16X2 : DT;  // dtype code of the tensors the convolutions consume (h2 / h2q)
    // fp16x2m: d(conv output) and the data gradients are plain fp16 (in buffers carved for 4 bytes per element)
    const int GT = mixed ? MN_F16 : AT, GD = mixed ? MN_F16 : DT;
    if (n == "p0") return give(p0, n1, AT);
    if (n == "gp0") return give(gp0, n1, GD);
    if (n == "pooled") return give(pooled, (long)B * 512, MN_F32);
    if (n == "feat") return give(feat, (long)B * cfg.feat_dim, MN_F32);
    if (n == "poses") return give(poses, (long)B * 6, MN_F32);
    if (n == "dposes") return give(dposes, (long)B * 6, MN_F32);
    if (n == "dz") return give(dz, (long)B * cfg.feat_dim, MN_F32);
    if (n == "dpooled") return give(dpooled, (long)B * 512, MN_F32);
    if (n == "dropmask") return give(dropmask, (long)B * cfg.feat_dim, MN_F32);
    if (n.size() > 2 && n[0] == 'b') {  // "b<block>.<tensor>", blocks numbered 0..15 in network order
      const size_t dot = n.find('.');
      if (dot != std::string::npos) {
        const int bi = atoi(n.substr(1, dot - 1).c_str());
        const std::string t = n.substr(dot + 1);
        if (bi >= 0 && bi < (int)blocks.size()) {
          Block& k = blocks[bi];
          const long no = k.u2.M * k.u2.cp.cout;
          if (t == "y1") return give(k.u1.y, no, DT);
          if (t == "a1") return give(k.a1, no, AT);
          if (t == "y2") return give(k.u2.y, no, DT);
          if (t == "out") return give(k.out, no, AT);
          if (t == "gy1") return give(k.u1.gy, no, GT);
          if (t == "ga1") return give(k.ga1, no, GD);
          if (t == "gy2") return give(k.u2.gy, no, GT);
          if (t == "gout") return give(k.gout, no, GD);
          if (k.down && t == "yd") return give(k.ud.y, no, DT);
          if (k.down && t == "zd") return give(k.zd, no, AT);
          if (k.down && t == "gyd") return give(k.ud.gy, no, GT);
        }
      }
    }
    return fail("mn_debug_tensor: unknown tensor '" + n + "'");
  }

  // ---- optimiser -----------------------------------------------------------------------------------
  // host-side effects of an optimiser step
  void after_optim_host() override {
    step += 1;
    weights_dirty = true;
  }
  int sync_step_to_device(hipStream_t s) override {
    long long v = step;
    hipMemcpyAsync(step_dev, &v, sizeof(v), hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    return check_launch("sync_step");
  }
  int64_t applied_steps() override {
    long long v = 0;
    hipDeviceSynchronize();
    if (hipMemcpy(&v, step_dev, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
  }
  int optim_step(float grad_mul, hipStream_t s) override {
    // squared gradient norm: for clip_grad_norm, and (fp16) as the overflow detector of this step
    if (max_grad_norm > 0.f || overflow_guard) {
      const int nb = sqnorm_grid(L.model_floats);
      if (deterministic && nb <= kSqPartials) {
        hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(nb), dim3(256), 0, s, (const float*)grads, (long)L.model_floats, sqnorm,
                           sq_partials);
        hipLaunchKernelGGL(sqnorm_fold_kernel, dim3(1), dim3(256), 0, s, (const double*)sq_partials, nb, sqnorm);
      } else {
        if (!sqnorm_clean) hipMemsetAsync(sqnorm, 0, sizeof(double), s);  // (a second optimiser step on one forward pass)
        sqnorm_clean = false;
        hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(nb), dim3(256), 0, s, (const float*)grads, (long)L.model_floats, sqnorm,
                           (double*)nullptr);
      }
    }
    attempts += 1;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, step_dev, beta1, beta2, bc_dev, (const double*)sqnorm,
                       overflow_guard ? overflow_dev : (long long*)nullptr, (long long)attempts);
    AdamArgs a;
    a.p = params; a.g = grads; a.m = m1; a.v = m2; a.n = L.param_floats; a.n_clip = L.model_floats; a.lr = lr; a.wd = wd;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.bc1 = 1.f; a.bc2 = 1.f; a.bc_dev = bc_dev;
    a.grad_mul = grad_mul; a.max_norm = max_grad_norm; a.sqnorm = sqnorm; a.frozen = frozen; a.eps_mode = cfg.eps_mode;
    a.skip = overflow_guard ? overflow_dev : nullptr;
    a.method = optim_method; a.nesterov = nesterov;
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(L.param_floats)), dim3(256), 0, s, a);
    if (overflow_guard && overflow_host)
      hipMemcpyAsync(overflow_host, overflow_dev, 4 * sizeof(long long), hipMemcpyDeviceToHost, s);
    return check_launch("optim_step");
  }
};

}  // namespace

struct mn_handle {
  std::unique_ptr<PlanBase> plan;
};

static int validate(const mn_config* c) {
  if (!c) return fail("null config");
  if (c->mode < 0 || c->mode > 3) return fail("config: bad mode");
  if (c->dtype != MN_DTYPE_F32 && c->dtype != MN_DTYPE_F16 && c->dtype != MN_DTYPE_F32X3 && c->dtype != MN_DTYPE_F16X2 &&
      c->dtype != MN_DTYPE_F16X2M && c->dtype != MN_DTYPE_F16X2Q)
    return fail("config: bad dtype");
  if (c->windows < 1 || c->T < 1 || c->T > kMaxT) return fail("config: windows >= 1 and 1 <= T <= 8 required");
  if (c->mode == MN_MODE_POSENET && c->T != 1) return fail("config: PoseNet mode requires T = 1");
  if (c->mode >= MN_MODE_MAPNET && c->T < 2) return fail("config: MapNet modes require T >= 2");
  if (c->H < 32 || c->W < 32) return fail("config: image must be at least 32x32");
  if (c->feat_dim < 64 || c->feat_dim % 64 != 0) return fail("config: feat_dim must be a multiple of 64");
  if (!(c->loss_scale > 0.f)) return fail("config: loss_scale must be positive");
  {  // the largest activation (stem output, NHWC) must stay below the 4 GiB a 32-bit buffer offset spans
    const long frames = c->mode == MN_MODE_POSENET ? 1 : (c->mode == MN_MODE_MAPNET ? c->T : 2 * c->T);
    const long px = ((c->H - 1) / 2 + 1) * (long)((c->W - 1) / 2 + 1);
    if (c->windows * frames * px * 64 * (c->dtype == MN_DTYPE_F16 ? 2 : 4) >= 0xfffffff0l)
      return fail("config: batch too large (an activation tensor would exceed 4 GiB); split the batch");
  }
  return 0;
}

extern "C" int64_t mn_plan_bytes(const mn_config* cfg) {
  if (validate(cfg)) return -1;
  if (cfg->dtype == MN_DTYPE_F16) {
    Plan<half> p(*cfg);
    return (int64_t)p.carve(nullptr);
  }
  Plan<float> p(*cfg);
  return (int64_t)p.carve(nullptr);
}

extern "C" mn_handle* mn_create(const mn_config* cfg, float* params, float* opt_state, void* buffers, void* work,
                                void* stream) {
  if (validate(cfg)) return nullptr;
  if (!params || !opt_state || !buffers || !work) {
    fail("mn_create: null arena");
    return nullptr;
  }
  mn_handle* h = new mn_handle();
  begin_call();
  int rc;
  if (cfg->dtype == MN_DTYPE_F16) {
    auto* p = new Plan<half>(*cfg);
    h->plan.reset(p);
    rc = p->attach(params, opt_state, buffers, work, (hipStream_t)stream);
  } else {
    auto* p = new Plan<float>(*cfg);
    h->plan.reset(p);
    rc = p->attach(params, opt_state, buffers, work, (hipStream_t)stream);
  }
  if (rc) {
    delete h;
    return nullptr;
  }
  return h;
}
extern "C" void mn_destroy(mn_handle* h) { delete h; }

#define MN_H(h)                                       \
  if (!(h) || !(h)->plan) return fail("null handle"); \
  begin_call();                                       \
  PlanBase& P = *(h)->plan;

extern "C" int mn_set_learn_flags(mn_handle* h, int learn_beta, int learn_gamma) {
  MN_H(h);
  P.learn_beta = learn_beta;
  P.learn_gamma = learn_gamma;
  if (P.cfg.dtype == MN_DTYPE_F16)
    static_cast<Plan<half>&>(P).update_frozen(nullptr);
  else
    static_cast<Plan<float>&>(P).update_frozen(nullptr);
  return 0;
}
extern "C" int mn_set_optim(mn_handle* h, float lr, float weight_decay, float beta1, float beta2, float eps,
                            float max_grad_norm) {
  MN_H(h);
  P.lr = lr; P.wd = weight_decay; P.beta1 = beta1; P.beta2 = beta2; P.eps = eps; P.max_grad_norm = max_grad_norm;
  return 0;
}
extern "C" int mn_set_optim_method(mn_handle* h, int method, int nesterov) {
  MN_H(h);
  if (method < 0 || method > 2) return fail("set_optim_method: method must be 0 (adam), 1 (sgd) or 2 (rmsprop)");
  P.optim_method = method;
  P.nesterov = nesterov ? 1 : 0;
  return 0;
}
extern "C" int mn_set_step_count(mn_handle* h, int64_t step) {
  MN_H(h);
  // steps still in flight on the caller's (non-blocking) stream increment the device counter in adam_prep_kernel: let them
  // finish before the counter is overwritten through the NULL stream
  if (hipDeviceSynchronize() != hipSuccess) return check_launch("set_step_count");
  P.step = step;
  return P.sync_step_to_device(nullptr);
}
extern "C" int64_t mn_get_step_count(mn_handle* h) { return (h && h->plan) ? h->plan->applied_steps() : -1; }
extern "C" int mn_set_loss_host(mn_handle* h, float* pinned_host) {
  MN_H(h);
  P.loss_host = pinned_host;
  P.loss_pending = false;
  return 0;
}
extern "C" int mn_wait_loss(mn_handle* h) {
  if (!h || !h->plan) return -1;
  PlanBase& P = *h->plan;
  if (!P.loss_pending) return 1;  // nothing was posted (default stream): read the device scalar instead
  P.loss_pending = false;
  if (hipEventSynchronize(P.loss_event) != hipSuccess) {
    (void)hipGetLastError();
    return 1;
  }
  return 0;
}
extern "C" int mn_debug_tensor(mn_handle* h, const char* name, void** ptr, int64_t* numel, int32_t* dtype) {
  MN_H(h);
  if (!name || !ptr || !numel || !dtype) return fail("mn_debug_tensor: null argument");
  return P.debug_tensor(name, ptr, numel, dtype);
}
extern "C" int mn_get_loss_scale(mn_handle* h, float* scale, int64_t* skipped_steps) {
  MN_H(h);
  if (scale) *scale = P.cur_scale;
  if (skipped_steps) *skipped_steps = P.overflow_host ? (int64_t) * (volatile long long*)(P.overflow_host + 1) : 0;
  return 0;
}
extern "C" int64_t mn_stuck_overflow_steps(mn_handle* h) { return (h && h->plan) ? h->plan->stuck_skips : -1; }
extern "C" int mn_set_loss_scale(mn_handle* h, float scale, int growth_interval) {
  MN_H(h);
  if (!(scale > 0.f)) return fail("mn_set_loss_scale: scale must be positive");
  if (P.cfg.dtype != MN_DTYPE_F16 && P.cfg.dtype != MN_DTYPE_F16X2 && P.cfg.dtype != MN_DTYPE_F16X2M &&
      P.cfg.dtype != MN_DTYPE_F16X2Q && scale != 1.f)
    return fail("mn_set_loss_scale: fp32 plans do not scale the loss");
  P.cur_scale = scale;
  P.scale_set_at = P.attempts;
  P.scale_growth_interval = growth_interval;
  P.clean_steps = 0;
  P.stuck_skips = 0;
  if (P.overflow_host) {  // skips of steps still in flight belong to the old scale: do not halve the new one for them
    hipDeviceSynchronize();
    P.skipped_seen = *(volatile long long*)(P.overflow_host + 1);
    P.done_seen = *(volatile long long*)(P.overflow_host + 3);
  }
  return 0;
}
extern "C" int mn_set_dropout(mn_handle* h, float p, uint64_t seed) {
  MN_H(h);
  if (!(p >= 0.f && p < 1.f)) return fail("mn_set_dropout: 0 <= p < 1 required");
  P.drop_p = p;
  P.drop_seed = seed;
  P.drop_calls = 0;
  return 0;
}
extern "C" int mn_set_dropout_calls(mn_handle* h, uint32_t calls) {
  MN_H(h);
  P.drop_calls = calls;
  return 0;
}
extern "C" int mn_set_input_u8(mn_handle* h, int enable, const float* mean, const float* std) {
  MN_H(h);
  if (enable && (!mean || !std)) return fail("mn_set_input_u8: mean and std (3 floats each, host memory) are required");
  if (enable)
    for (int c = 0; c < 3; ++c) {
      if (!(std[c] > 0.f)) return fail("mn_set_input_u8: std must be positive");
      P.input_norm.scale[c] = 1.f / (255.f * std[c]);
      P.input_norm.shift[c] = -mean[c] / std[c];
    }
  P.input_u8 = enable != 0;
  return 0;
}
extern "C" int mn_forward(mn_handle* h, const void* images, float* poses_out, int training, void* stream) {
  MN_H(h);
  return P.forward(images, poses_out, training, (hipStream_t)stream);
}
extern "C" int mn_loss(mn_handle* h, const float* pred, const float* targ, float* loss_out, void* stream) {
  MN_H(h);
  return P.loss_only(pred, targ, loss_out, (hipStream_t)stream);
}
extern "C" int mn_train_forward_loss(mn_handle* h, const void* images, const float* targets, float* loss_out,
                                     float* poses_out, void* stream) {
  MN_H(h);
  P.timer.reset();
  hipStream_t s = (hipStream_t)stream;
  P.weights_dirty = true;  // a training forward always follows an optimiser step or a parameter load
  P.poll_overflow();
  return P.forward_loss(images, targets, loss_out, poses_out, s);
}
extern "C" int mn_train_backward_stage(mn_handle* h, int stage, void* stream) {
  MN_H(h);
  if (stage < 0 || stage > 3) return fail("backward_stage: stage must be 0..3");
  hipStream_t s = (hipStream_t)stream;
  return P.backward_stage(stage, s);
}
extern "C" int mn_grad_bucket(mn_handle* h, int stage, int64_t* offset, int64_t* count) {
  MN_H(h);
  if (stage < 0 || stage > 3) return fail("mn_grad_bucket: stage must be 0..3");
  *offset = P.L.stage_begin[stage];
  *count = P.L.stage_end[stage] - P.L.stage_begin[stage];
  return 0;
}
extern "C" int mn_grad_bucket_pack_bf16(mn_handle* h, int stage, void* out_bf16, void* stream) {
  MN_H(h);
  if (stage < 0 || stage > 3 || !out_bf16) return fail("mn_grad_bucket_pack_bf16: stage 0..3 and an output buffer are required");
  const long off = P.L.stage_begin[stage], n = P.L.stage_end[stage] - off;
  hipLaunchKernelGGL(grad_pack_bf16_kernel, dim3(grad_transport_grid(n)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)(P.grad_arena() + off), (__bf16*)out_bf16, n);
  return check_launch("grad_bucket_pack_bf16");
}
extern "C" int mn_grad_bucket_unpack_bf16(mn_handle* h, int stage, const void* in_bf16, void* stream) {
  MN_H(h);
  if (stage < 0 || stage > 3 || !in_bf16) return fail("mn_grad_bucket_unpack_bf16: stage 0..3 and an input buffer are required");
  const long off = P.L.stage_begin[stage], n = P.L.stage_end[stage] - off;
  hipLaunchKernelGGL(grad_unpack_bf16_kernel, dim3(grad_transport_grid(n)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)in_bf16,
                     P.grad_arena() + off, n);
  return check_launch("grad_bucket_unpack_bf16");
}
extern "C" int mn_optim_step(mn_handle* h, float grad_mul, void* stream) {
  MN_H(h);
  hipStream_t s = (hipStream_t)stream;
  int rc = P.optim_step(grad_mul, s);
  if (rc == 0) P.after_optim_host();
  return rc;
}
extern "C" int mn_train_step(mn_handle* h, const void* images, const float* targets, float* loss_out, float* poses_out,
                             void* stream) {
  MN_H(h);
  hipStream_t s = (hipStream_t)stream;
  P.timer.reset();
  P.weights_dirty = true;  // a training step always follows an optimiser step or a parameter load
  P.poll_overflow();
  auto* tp = P.timer.begin(3, s);
  int rc = P.forward_loss(images, targets, loss_out, poses_out, s);
  for (int st = 3; st >= 0 && rc == 0; --st) rc = P.backward_stage(st, s);
  if (rc == 0) rc = P.optim_step(1.f, s);
  P.timer.end(tp, s);
  if (rc == 0) P.after_optim_host();
  return rc;
}
extern "C" int mn_params_changed(mn_handle* h) {
  MN_H(h);
  P.weights_dirty = true;
  return 0;
}
extern "C" int mn_set_profiling(mn_handle* h, int enable) {
  MN_H(h);
  P.timer.enabled = enable != 0;
  P.timer.reset();
  return 0;
}
extern "C" int mn_last_kernel_ms(mn_handle* h, int which, float* ms, int* launches) {
  MN_H(h);
  if (which < 0 || which > 3) return fail("mn_last_kernel_ms: which must be 0..3");
  P.timer.collect();
  if (which == 0) {
    *ms = P.timer.ms[0] + P.timer.ms[1];
    *launches = P.timer.launches[0] + P.timer.launches[1];
  } else if (which == 1) {
    *ms = P.timer.ms[0];
    *launches = P.timer.launches[0];
  } else if (which == 2) {
    *ms = P.timer.ms[1];
    *launches = P.timer.launches[1];
  } else {
    *ms = P.timer.ms[3];
    *launches = P.timer.launches[3];
  }
  return 0;
}
