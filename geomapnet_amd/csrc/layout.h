// Parameter / buffer arena layout of PoseNet-over-ResNet-34, in the reference's registration
// order with the reference's state_dict keys (torchvision ResNet naming under
// `feature_extractor.`, /root/reference/models/posenet.py:43-49).
#pragma once
#include <stdint.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mapnet_hip.h"

namespace mn {

struct ConvP {
  int64_t w;  // offset of the OHWI master weight in the parameter arena
  int cin, cout, k, stride, pad;
};
struct BnP {
  int64_t gamma, beta;  // parameter arena (floats)
  int64_t rm, rv;       // buffer arena, float index
  int64_t nbt;          // buffer arena, byte offset of the int64 counter
  int C;
};
struct BlockP {
  ConvP c1, c2, cd;
  BnP b1, b2, bd;
  bool down;
  int stage;
};

struct Layout {
  std::vector<mn_entry> entries;
  int64_t param_floats = 0;  // including the 4 criterion scalars at the tail
  int64_t model_floats = 0;  // model parameters only (clip_grad_norm range)
  int64_t buffer_bytes = 0;
  ConvP stem;
  BnP stem_bn;
  std::vector<BlockP> blocks;
  int64_t fc_w, fc_b, xyz_w, xyz_b, wpqr_w, wpqr_b, crit;
  int feat_dim;
  int64_t stage_begin[4], stage_end[4];

  explicit Layout(int feat) : feat_dim(feat) { build(); }

 private:
  int64_t pcur = 0, bcur = 0;
  int cur_stage = 0;

  void add(const std::string& name, int64_t off, int64_t numel, std::vector<int> shape, int is_buffer, int is_int64,
           int ohwi) {
    mn_entry e;
    memset(&e, 0, sizeof(e));
    snprintf(e.name, sizeof(e.name), "%s", name.c_str());
    e.offset = off;
    e.numel = numel;
    e.ndim = (int)shape.size();
    for (size_t i = 0; i < shape.size() && i < 4; ++i) e.shape[i] = shape[i];
    e.is_buffer = is_buffer;
    e.is_int64 = is_int64;
    e.ohwi = ohwi;
    e.stage = cur_stage;
    entries.push_back(e);
  }
  ConvP conv(const std::string& name, int cin, int cout, int k, int stride, int pad) {
    ConvP c{pcur, cin, cout, k, stride, pad};
    int64_t n = (int64_t)cout * cin * k * k;
    add(name + ".weight", pcur, n, {cout, cin, k, k}, 0, 0, 1);
    pcur += n;
    return c;
  }
  BnP bn(const std::string& name, int C) {
    BnP b;
    b.C = C;
    b.gamma = pcur;
    add(name + ".weight", pcur, C, {C}, 0, 0, 0);
    pcur += C;
    b.beta = pcur;
    add(name + ".bias", pcur, C, {C}, 0, 0, 0);
    pcur += C;
    b.rm = bcur / 4;
    add(name + ".running_mean", bcur / 4, C, {C}, 1, 0, 0);
    bcur += (int64_t)C * 4;
    b.rv = bcur / 4;
    add(name + ".running_var", bcur / 4, C, {C}, 1, 0, 0);
    bcur += (int64_t)C * 4;
    b.nbt = bcur;
    add(name + ".num_batches_tracked", bcur / 8, 1, {}, 1, 1, 0);
    bcur += 8;
    return b;
  }
  int64_t vec(const std::string& name, std::vector<int> shape) {
    int64_t n = 1;
    for (int s : shape) n *= s;
    int64_t off = pcur;
    add(name, pcur, n, shape, 0, 0, 0);
    pcur += n;
    return off;
  }
  void build() {
    const std::string fe = "feature_extractor.";
    cur_stage = 0;
    stage_begin[0] = 0;
    stem = conv(fe + "conv1", 3, 64, 7, 2, 3);
    stem_bn = bn(fe + "bn1", 64);
    const int widths[4] = {64, 128, 256, 512}, counts[4] = {3, 4, 6, 3};
    int cin = 64;
    for (int li = 0; li < 4; ++li) {
      if (li > 0) {
        stage_end[cur_stage] = pcur;
        cur_stage = li;
        stage_begin[cur_stage] = pcur;
      }
      for (int b = 0; b < counts[li]; ++b) {
        std::string p = fe + "layer" + std::to_string(li + 1) + "." + std::to_string(b) + ".";
        int stride = (b == 0 && li > 0) ? 2 : 1;
        BlockP blk;
        blk.stage = cur_stage;
        blk.down = (stride != 1 || cin != widths[li]);
        blk.c1 = conv(p + "conv1", cin, widths[li], 3, stride, 1);
        blk.b1 = bn(p + "bn1", widths[li]);
        blk.c2 = conv(p + "conv2", widths[li], widths[li], 3, 1, 1);
        blk.b2 = bn(p + "bn2", widths[li]);
        if (blk.down) {
          blk.cd = conv(p + "downsample.0", cin, widths[li], 1, stride, 0);
          blk.bd = bn(p + "downsample.1", widths[li]);
        }
        blocks.push_back(blk);
        cin = widths[li];
      }
    }
    fc_w = vec(fe + "fc.weight", {feat_dim, 512});
    fc_b = vec(fe + "fc.bias", {feat_dim});
    xyz_w = vec("fc_xyz.weight", {3, feat_dim});
    xyz_b = vec("fc_xyz.bias", {3});
    wpqr_w = vec("fc_wpqr.weight", {3, feat_dim});
    wpqr_b = vec("fc_wpqr.bias", {3});
    model_floats = pcur;
    crit = pcur;
    pcur += 4;  // sax, saq, srx, srq (owned by the criterion module on the Python side)
    stage_end[3] = pcur;
    param_floats = pcur;
    buffer_bytes = bcur;
  }
};

}  // namespace mn
