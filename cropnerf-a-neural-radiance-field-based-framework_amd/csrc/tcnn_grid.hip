// tcnn_grid.hip -- tiny-cuda-nn-compatible hash grids: geometry ("plan"), conversion between tcnn's packed parameter
// vector and this library's table, and the parameter tying that training on such a table needs.
//
// The reference builds FruitField with implementation="tcnn" (fruit_nerf/fruit_field.py:95,125-132): its checkpoints
// hold `field.mlp_base_grid.tcnn_encoding.params` / `proposal_networks.N.mlp_base.tcnn_encoding.params` in tcnn's
// GridEncoding layout (tiny-cuda-nn encodings/grid.h: grid_scale, grid_resolution, grid_index and the offset table of
// GridEncodingTemplated -- restated here from the published source; tcnn itself is not in /root/reference).
//
// Layout of this library's table for such a grid (cn_grid.layout = CN_GRID_TCNN):
//   * hashed level (res^3 > 2^log2_T): 2^log2_T entries, entry = (x ^ y*2654435761 ^ z*805459861) mod 2^log2_T -- as tcnn;
//   * dense level: tcnn stores entry x + y*res + z*res^2 (padded to a multiple of 8).  Here the level has power-of-two
//     strides, entry = x | y << b | z << 2b with b = ceil(log2(res + 1)), so that the kernels index hashed and dense levels
//     with ONE expression (xor of three per-axis terms, cn_common.hpp: Lvl) instead of a per-level branch in their
//     innermost loop; the price is (2^b / res)^3 times the memory of the few coarse levels (a few MB).
//   * tcnn does not clamp: a sample in the upper half-cell of an axis has the corner index `res`, which in tcnn's linear
//     array is the first element of the next row (and (res, res-1, res-1) wraps to entry res^3 mod size).  cn_tcnn_grid_pack
//     reproduces that by filling every entry (x, y, z) in [0, res]^3 with the tcnn parameter at (x + y*res + z*res^2) mod size:
//     several entries may alias one parameter; cn_tcnn_grid_tie_* keep them consistent during training.
#include "cn_common.hpp"
#include "cn_det.hpp"

#include <cmath>
#include <type_traits>

namespace cn {

struct PlanDev {
  int num_levels;
  unsigned res[CN_MAX_LEVELS];
  unsigned size[CN_MAX_LEVELS];     // tcnn entries of the level (hashmap_size)
  unsigned packed[CN_MAX_LEVELS];   // tcnn offset, entries
  unsigned off[CN_MAX_LEVELS + 1];  // this library's offset, entries
  unsigned bits[CN_MAX_LEVELS];
};

static PlanDev make_plan_dev(const cn_tcnn_grid_plan& p) {
  PlanDev d{};
  d.num_levels = p.num_levels;
  for (int l = 0; l < p.num_levels; ++l) {
    d.res[l] = p.resolution[l];
    d.size[l] = p.packed_offset[l + 1] - p.packed_offset[l];
    d.packed[l] = p.packed_offset[l];
    d.off[l] = p.level_offset[l];
    d.bits[l] = p.level_bits[l];
  }
  d.off[p.num_levels] = p.level_offset[p.num_levels];
  return d;
}

// level of a table entry (levels are few: linear scan over the kernarg offsets, wave-divergent only at level borders)
__device__ __forceinline__ int level_of(const unsigned* off, int n, unsigned e) {
  int l = 0;
#pragma unroll 1
  for (int k = 1; k < n; ++k) l = e >= off[k] ? k : l;
  return l;
}

// tcnn parameter index (within the level) an entry of a dense level stands for, or -1 for an unreachable entry
__device__ __forceinline__ long long dense_param_of(unsigned local, unsigned b, unsigned res, unsigned size) {
  const unsigned m = (1u << b) - 1u;
  const unsigned x = local & m, y = (local >> b) & m, z = local >> (2 * b);
  if (x > res || y > res || z > res) return -1;
  const unsigned long long lin = (unsigned long long)x + (unsigned long long)y * res + (unsigned long long)z * res * res;
  return (long long)(lin % size);
}

// the entry of this library's table that OWNS tcnn parameter `p` of a dense level (its canonical coordinates)
__device__ __forceinline__ unsigned dense_owner_of(unsigned p, unsigned b, unsigned res) {
  const unsigned x = p % res, y = (p / res) % res, z = p / (res * res);  // z == res for the padding entries
  return x | (y << b) | (z << (2 * b));
}

template <typename T>
__device__ __forceinline__ float2 load2(const T* base, unsigned long long entry) {
  return make_float2((float)base[2 * entry], (float)base[2 * entry + 1]);
}
template <typename T>
__device__ __forceinline__ void store2(T* base, unsigned long long entry, float2 v) {
  base[2 * entry] = (T)v.x;
  base[2 * entry + 1] = (T)v.y;
}

template <typename TP, typename TT>
__global__ void __launch_bounds__(256) tcnn_pack_kernel(PlanDev P, const TP* __restrict__ packed, TT* __restrict__ table) {
  const unsigned total = P.off[P.num_levels];
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    const int l = level_of(P.off, P.num_levels, e);
    const unsigned local = e - P.off[l];
    float2 v = make_float2(0.f, 0.f);
    if (P.bits[l] == 0) {
      v = load2(packed, (unsigned long long)P.packed[l] + local);
    } else {
      const long long p = dense_param_of(local, P.bits[l], P.res[l], P.size[l]);
      if (p >= 0) v = load2(packed, (unsigned long long)P.packed[l] + (unsigned long long)p);
    }
    store2(table, e, v);
  }
}

template <typename TP, typename TT>
__global__ void __launch_bounds__(256) tcnn_unpack_kernel(PlanDev P, const TT* __restrict__ table, TP* __restrict__ packed,
                                                          unsigned total_packed) {
  for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < total_packed; q += gridDim.x * 256u) {
    int l = 0;
#pragma unroll 1
    for (int k = 1; k < P.num_levels; ++k) l = q >= P.packed[k] ? k : l;
    const unsigned p = q - P.packed[l];
    const unsigned local = P.bits[l] == 0 ? p : dense_owner_of(p, P.bits[l], P.res[l]);
    store2(packed, q, load2(table, (unsigned long long)P.off[l] + local));
  }
}

// FOLD: grad[owner] += grad[alias]; grad[alias] = 0.   !FOLD: table[alias] = table[owner].
template <bool FOLD>
__global__ void __launch_bounds__(256) tcnn_tie_kernel(PlanDev P, float* __restrict__ table) {
  const unsigned total = P.off[P.num_levels];
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    const int l = level_of(P.off, P.num_levels, e);
    if (P.bits[l] == 0) continue;
    const unsigned local = e - P.off[l];
    const long long p = dense_param_of(local, P.bits[l], P.res[l], P.size[l]);
    if (p < 0) continue;
    const unsigned owner = dense_owner_of((unsigned)p, P.bits[l], P.res[l]);
    if (owner == local) continue;
    float* a = table + 2ull * e;
    float* o = table + 2ull * ((unsigned long long)P.off[l] + owner);
    if (FOLD) {
      const float g0 = a[0], g1 = a[1];
      if (g0 != 0.f) cn_atomic_add(o, g0);
      if (g1 != 0.f) cn_atomic_add(o + 1, g1);
      a[0] = 0.f;
      a[1] = 0.f;
    } else {
      a[0] = o[0];
      a[1] = o[1];
    }
  }
}

static int check_plan(const cn_tcnn_grid_plan* plan, const char* who) {
  CN_REQUIRE(plan, CN_ERR_INVALID, "%s: null plan", who);
  CN_REQUIRE(plan->num_levels >= 1 && plan->num_levels <= CN_MAX_LEVELS, CN_ERR_INVALID, "%s: plan with %d levels", who,
             plan->num_levels);
  CN_REQUIRE(plan->level_offset[plan->num_levels] > 0, CN_ERR_INVALID, "%s: plan not initialised", who);
  return CN_OK;
}

}  // namespace cn

extern "C" int cn_tcnn_grid_plan_init(int32_t num_levels, int32_t log2_table_size, int32_t base_resolution,
                                      float per_level_scale, cn_tcnn_grid_plan* plan) {
  CN_REQUIRE(plan, CN_ERR_INVALID, "cn_tcnn_grid_plan_init: null plan");
  CN_REQUIRE(num_levels >= 1 && num_levels <= CN_MAX_LEVELS, CN_ERR_UNSUPPORTED, "cn_tcnn_grid_plan_init: %d levels",
             num_levels);
  CN_REQUIRE(log2_table_size >= 3 && log2_table_size <= 24, CN_ERR_UNSUPPORTED,
             "cn_tcnn_grid_plan_init: log2 table size %d", log2_table_size);
  CN_REQUIRE(base_resolution >= 1 && per_level_scale >= 1.f, CN_ERR_INVALID,
             "cn_tcnn_grid_plan_init: base resolution %d, per-level scale %g", base_resolution, (double)per_level_scale);
  std::memset(plan, 0, sizeof(*plan));
  plan->num_levels = num_levels;
  plan->log2_table_size = log2_table_size;
  plan->base_resolution = base_resolution;
  plan->per_level_scale = per_level_scale;
  const float log2_scale = std::log2(per_level_scale);  // float overload, as in tcnn's constructor
  const uint32_t T = 1u << log2_table_size;
  uint64_t packed = 0, off = 0;
  for (int l = 0; l < num_levels; ++l) {
    // grid.h: grid_scale = exp2f(level * log2_per_level_scale) * base_resolution - 1; grid_resolution = ceilf(scale) + 1
    const float scale = exp2f((float)l * log2_scale) * (float)base_resolution - 1.0f;
    const uint32_t res = (uint32_t)ceilf(scale) + 1u;
    plan->scalings[l] = scale;
    plan->resolution[l] = res;
    // GridEncodingTemplated: params_in_level = min(next_multiple(res^3 (capped at 2^31-1), 8), 2^log2_T)
    const uint64_t max_params = 0xffffffffull / 2;
    uint64_t dense = (double)res * res * res > (double)max_params ? max_params : (uint64_t)res * res * res;
    dense = (dense + 7) / 8 * 8;
    const uint64_t n = dense < T ? dense : T;
    // grid_index: hashed iff the dense stride product exceeds the level's parameter count
    const bool hashed = (uint64_t)res * res * res > n;
    int b = 0;
    if (!hashed) {
      while ((1u << b) < res + 1u) ++b;  // coordinates 0..res inclusive
      CN_REQUIRE(b <= 9, CN_ERR_UNSUPPORTED, "cn_tcnn_grid_plan_init: dense level %d of resolution %u", l, res);
    }
    plan->level_bits[l] = (uint8_t)b;
    plan->packed_offset[l] = (uint32_t)packed;
    plan->level_offset[l] = (uint32_t)off;
    packed += n;
    off += hashed ? T : (1ull << (3 * b));
    CN_REQUIRE(off * 8ull <= (1ull << 31), CN_ERR_UNSUPPORTED, "cn_tcnn_grid_plan_init: table larger than 2 GiB");
  }
  plan->packed_offset[num_levels] = (uint32_t)packed;
  plan->level_offset[num_levels] = (uint32_t)off;
  return CN_OK;
}

extern "C" int cn_tcnn_grid_describe(const cn_tcnn_grid_plan* plan, const void* table, int32_t table_dtype,
                                     cn_grid* grid) {
  int rc = cn::check_plan(plan, "cn_tcnn_grid_describe");
  if (rc) return rc;
  CN_REQUIRE(grid, CN_ERR_INVALID, "cn_tcnn_grid_describe: null grid");
  CN_REQUIRE(table_dtype == CN_TABLE_F32 || table_dtype == CN_TABLE_F16, CN_ERR_INVALID,
             "cn_tcnn_grid_describe: table dtype %d", table_dtype);
  std::memset(grid, 0, sizeof(*grid));
  grid->table = table;
  grid->num_levels = plan->num_levels;
  grid->log2_table_size = plan->log2_table_size;
  grid->layout = CN_GRID_TCNN;
  grid->table_dtype = table_dtype;
  for (int l = 0; l < plan->num_levels; ++l) {
    grid->scalings[l] = plan->scalings[l];
    grid->level_offset[l] = plan->level_offset[l];
    grid->level_bits[l] = plan->level_bits[l];
  }
  return CN_OK;
}

namespace {
template <typename F>
int dispatch2(int32_t a, int32_t b, F&& f) {  // (packed dtype, table dtype) -> typed call
  if (a == CN_TABLE_F32 && b == CN_TABLE_F32) return f((float*)nullptr, (float*)nullptr);
  if (a == CN_TABLE_F32 && b == CN_TABLE_F16) return f((float*)nullptr, (_Float16*)nullptr);
  if (a == CN_TABLE_F16 && b == CN_TABLE_F32) return f((_Float16*)nullptr, (float*)nullptr);
  if (a == CN_TABLE_F16 && b == CN_TABLE_F16) return f((_Float16*)nullptr, (_Float16*)nullptr);
  cn::set_error("cn_tcnn_grid: dtype %d / %d", a, b);
  return CN_ERR_INVALID;
}
}  // namespace

extern "C" int cn_tcnn_grid_pack(const cn_tcnn_grid_plan* plan, const void* packed, int32_t packed_dtype, void* table,
                                 int32_t table_dtype, cn_stream_t stream) {
  int rc = cn::check_plan(plan, "cn_tcnn_grid_pack");
  if (rc) return rc;
  CN_REQUIRE(packed && table, CN_ERR_INVALID, "cn_tcnn_grid_pack: null buffer");
  const cn::PlanDev P = cn::make_plan_dev(*plan);
  const unsigned total = P.off[P.num_levels];
  rc = dispatch2(packed_dtype, table_dtype, [&](auto* tp, auto* tt) {
    using TP = std::remove_pointer_t<decltype(tp)>;
    using TT = std::remove_pointer_t<decltype(tt)>;
    hipLaunchKernelGGL((cn::tcnn_pack_kernel<TP, TT>), dim3(cn::grid_for(total, 256, 256 * 16)), dim3(256), 0,
                       cn::as_stream(stream), P, static_cast<const TP*>(packed), static_cast<TT*>(table));
    return CN_OK;
  });
  return rc ? rc : cn::check_launch("cn_tcnn_grid_pack");
}

extern "C" int cn_tcnn_grid_unpack(const cn_tcnn_grid_plan* plan, const void* table, int32_t table_dtype, void* packed,
                                   int32_t packed_dtype, cn_stream_t stream) {
  int rc = cn::check_plan(plan, "cn_tcnn_grid_unpack");
  if (rc) return rc;
  CN_REQUIRE(packed && table, CN_ERR_INVALID, "cn_tcnn_grid_unpack: null buffer");
  const cn::PlanDev P = cn::make_plan_dev(*plan);
  const unsigned total = plan->packed_offset[plan->num_levels];
  rc = dispatch2(packed_dtype, table_dtype, [&](auto* tp, auto* tt) {
    using TP = std::remove_pointer_t<decltype(tp)>;
    using TT = std::remove_pointer_t<decltype(tt)>;
    hipLaunchKernelGGL((cn::tcnn_unpack_kernel<TP, TT>), dim3(cn::grid_for(total, 256, 256 * 16)), dim3(256), 0,
                       cn::as_stream(stream), P, static_cast<const TT*>(table), static_cast<TP*>(packed), total);
    return CN_OK;
  });
  return rc ? rc : cn::check_launch("cn_tcnn_grid_unpack");
}

extern "C" int cn_tcnn_grid_tie_gradients(const cn_tcnn_grid_plan* plan, float* grad_table, cn_stream_t stream) {
  int rc = cn::check_plan(plan, "cn_tcnn_grid_tie_gradients");
  if (rc) return rc;
  CN_REQUIRE(grad_table, CN_ERR_INVALID, "cn_tcnn_grid_tie_gradients: null buffer");
  const cn::PlanDev P = cn::make_plan_dev(*plan);
  // only the dense levels (the first ones) hold aliases: stop the sweep at the end of the last dense level
  unsigned end = 0;
  for (int l = 0; l < P.num_levels; ++l)
    if (P.bits[l]) end = P.off[l + 1];
  if (!end) return CN_OK;
  cn::PlanDev Q = P;
  Q.off[Q.num_levels] = end;
  hipLaunchKernelGGL(cn::tcnn_tie_kernel<true>, dim3(cn::grid_for(end, 256, 256 * 8)), dim3(256), 0,
                     cn::as_stream(stream), Q, grad_table);
  if (int rc2 = cn::check_launch("cn_tcnn_grid_tie_gradients")) return rc2;
  CN_DET_FLUSH(cn::as_stream(stream));
  return CN_OK;
}

extern "C" int cn_tcnn_grid_tie_parameters(const cn_tcnn_grid_plan* plan, float* table, cn_stream_t stream) {
  int rc = cn::check_plan(plan, "cn_tcnn_grid_tie_parameters");
  if (rc) return rc;
  CN_REQUIRE(table, CN_ERR_INVALID, "cn_tcnn_grid_tie_parameters: null buffer");
  const cn::PlanDev P = cn::make_plan_dev(*plan);
  unsigned end = 0;
  for (int l = 0; l < P.num_levels; ++l)
    if (P.bits[l]) end = P.off[l + 1];
  if (!end) return CN_OK;
  cn::PlanDev Q = P;
  Q.off[Q.num_levels] = end;
  hipLaunchKernelGGL(cn::tcnn_tie_kernel<false>, dim3(cn::grid_for(end, 256, 256 * 8)), dim3(256), 0,
                     cn::as_stream(stream), Q, table);
  return cn::check_launch("cn_tcnn_grid_tie_parameters");
}
