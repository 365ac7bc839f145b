// Multi-tensor weight re-packing: every stale LDS-image layout of both networks in one launch (the optimizer step invalidates
// ~60 of them -- direct, Winograd, sub-pixel and stride-2 forms, forward and data-gradient variants -- and 60 launches of 5 us
// were 5 % of a level-4 step).  Records travel in the kernel-argument segment; blockIdx.y selects the record.
#include "pack_kernels.h"

namespace {

constexpr int PACK_CHUNK = 64;
struct PackChunk {
  mg_pack_desc_t d[PACK_CHUNK];
};

__global__ void __launch_bounds__(256) pack_multi_k(const PackChunk c) {
  const mg_pack_desc_t d = c.d[blockIdx.y];
  size_t total;
  switch (d.kind) {
    case MG_PACK_CONV3X3: total = pack_conv3x3_total(d.Co, d.Ci, d.dgrad); break;
    case MG_PACK_WINO3X3: total = pack_wino3x3_threads(d.Co, d.Ci, d.dgrad); break;
    case MG_PACK_UPCONV3X3: total = pack_upconv3x3_total(d.Co, d.Ci); break;
    case MG_PACK_SMALLNET: total = pack_smallnet_total(d.Co, d.Ci, d.dgrad); break;
    case MG_PACK_WINOUPS: total = pack_winoups_threads(d.Co, d.Ci, d.dgrad); break;
    default: total = pack_downconv_total(d.Co, d.Ci); break;
  }
  const int NT = pack_wino_nt_padded(d.dgrad ? d.Ci : d.Co);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    switch (d.kind) {
      case MG_PACK_CONV3X3: pack_conv3x3_elem(e, d.w, d.out, d.Co, d.Ci, d.dgrad); break;
      case MG_PACK_WINO3X3: pack_wino3x3_elem(e, d.w, d.out, d.Co, d.Ci, d.dgrad, NT); break;
      case MG_PACK_UPCONV3X3: pack_upconv3x3_elem(e, d.w, d.out, d.Co, d.Ci); break;
      case MG_PACK_SMALLNET: pack_smallnet_elem(e, d.w, d.out, d.Co, d.Ci, d.dgrad); break;
      case MG_PACK_WINOUPS: pack_winoups_elem(e, d.w, d.out, d.Co, d.Ci, d.dgrad); break;
      default: pack_downconv_elem(e, d.w, d.out, d.Co, d.Ci); break;
    }
  }
}

}  // namespace

extern "C" int mg_pack_multi(const mg_pack_desc_t* descs, int n, mg_stream_t stream) {
  MG_CHECK_ARG(descs && n > 0, "mg_pack_multi: bad arguments");
  for (int first = 0; first < n; first += PACK_CHUNK) {
    const int m = n - first < PACK_CHUNK ? n - first : PACK_CHUNK;
    PackChunk c;
    size_t most = 0;
    for (int i = 0; i < m; ++i) {
      const mg_pack_desc_t& d = descs[first + i];
      MG_CHECK_ARG(d.w && d.out && d.Co > 0 && d.Ci > 0 && d.kind >= MG_PACK_CONV3X3 && d.kind <= MG_PACK_WINOUPS,
                   "mg_pack_multi: bad record %d", first + i);
      c.d[i] = d;
      size_t total;
      switch (d.kind) {
        case MG_PACK_CONV3X3: total = pack_conv3x3_total(d.Co, d.Ci, d.dgrad); break;
        case MG_PACK_WINO3X3: total = pack_wino3x3_threads(d.Co, d.Ci, d.dgrad); break;
        case MG_PACK_UPCONV3X3: total = pack_upconv3x3_total(d.Co, d.Ci); break;
        case MG_PACK_SMALLNET: total = pack_smallnet_total(d.Co, d.Ci, d.dgrad); break;
        case MG_PACK_WINOUPS: total = pack_winoups_threads(d.Co, d.Ci, d.dgrad); break;
        default: total = pack_downconv_total(d.Co, d.Ci); break;
      }
      if (total > most) most = total;
    }
    size_t bx = (most + 255) / 256;
    if (bx > 256) bx = 256;  // x m records: thousands of workgroups, grid-stride over the larger layouts
    hipLaunchKernelGGL(pack_multi_k, dim3((unsigned)bx, (unsigned)m), dim3(256), 0, (hipStream_t)stream, c);
    MG_CHECK_LAUNCH("mg_pack_multi");
  }
  return MG_OK;
}
