// Replay storage kernels: coalesced transition gather and the closed-form
// position->id map of the reference's swap-remove list.
//
// Roofline: HBM bandwidth.  Algorithmic bytes per sampled transition =
// 2 x 28224 B read + 2 x 28224 B written.
#include "dz_common.h"

namespace {

struct GatherArgs {
  dz_field_t f[DZ_MAX_FIELDS];
  int num_fields;
};

// grid = (chunks, batch, fields); one 16-byte vector per lane => a wave moves
// 1 KiB per instruction, rows are contiguous so every access is coalesced.
__global__ __launch_bounds__(256) void gather_rows_kernel(
    GatherArgs a, const int64_t* __restrict__ ids, int64_t capacity) {
  const dz_field_t fd = a.f[blockIdx.z];
  const int b = blockIdx.y;
  const int64_t slot = dz_mod(ids[b], capacity);
  const int64_t rb = fd.row_bytes;
  const char* src = (const char*)fd.src + slot * rb;
  char* dst = (char*)fd.dst + (int64_t)b * rb;
  const bool vec_ok = ((rb & 15) == 0) && ((((uintptr_t)fd.src) & 15) == 0) &&
                      ((((uintptr_t)fd.dst) & 15) == 0);
  if (vec_ok) {
    const int64_t nvec = rb >> 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (int64_t)gridDim.x * blockDim.x) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 v = __builtin_nontemporal_load((const u32x4*)src + i);
      ((u32x4*)dst)[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rb;
         i += (int64_t)gridDim.x * blockDim.x) {
      dst[i] = src[i];
    }
  }
}

// ref: replay.py:52-82.  While filling, _ids[j] = j.  Once `t` >= capacity
// items have been added (one at a time, oldest evicted first):
//   _ids[N-1] = t-1;  _ids[j] = base + ((j - base) mod (N-1)), base = t - N.
__global__ void pos_to_id_kernel(const int64_t* __restrict__ pos, int n,
                                 int64_t t, int64_t size, int64_t capacity,
                                 int64_t* __restrict__ ids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t j = pos[i];
  int64_t id;
  if (t <= capacity || capacity == 1) {
    id = (capacity == 1) ? t - 1 : j;
  } else if (j == capacity - 1) {
    id = t - 1;
  } else {
    const int64_t base = t - capacity;
    id = base + dz_mod(j - base, capacity - 1);
  }
  ids[i] = id;
}

// Uniform sample in ONE launch: positions (host RNG draws, in the kernel
// arguments) -> ids by the closed form above -> row gather.  Block (0,b,0) also
// publishes ids[b].
constexpr int kMaxHostPos = 64;
struct HostPositions { int64_t pos[kMaxHostPos]; };
__device__ __forceinline__ int64_t uniform_id_at(int64_t j, int64_t t, int64_t capacity) {
  if (t <= capacity || capacity == 1) return (capacity == 1) ? t - 1 : j;
  if (j == capacity - 1) return t - 1;
  const int64_t base = t - capacity;
  return base + dz_mod(j - base, capacity - 1);
}
__global__ __launch_bounds__(256) void gather_uniform_kernel(
    GatherArgs a, HostPositions hp, int64_t t, int64_t capacity, int64_t* __restrict__ ids_out) {
  const dz_field_t fd = a.f[blockIdx.z];
  const int b = blockIdx.y;
  const int64_t id = uniform_id_at(hp.pos[b & (kMaxHostPos - 1)], t, capacity);
  if (ids_out && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) ids_out[b] = id;
  const int64_t slot = dz_mod(id, capacity);
  const int64_t rb = fd.row_bytes;
  const char* src = (const char*)fd.src + slot * rb;
  char* dst = (char*)fd.dst + (int64_t)b * rb;
  const bool vec_ok = ((rb & 15) == 0) && ((((uintptr_t)fd.src) & 15) == 0) &&
                      ((((uintptr_t)fd.dst) & 15) == 0);
  if (vec_ok) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int64_t nvec = rb >> 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (int64_t)gridDim.x * blockDim.x)
      ((u32x4*)dst)[i] = __builtin_nontemporal_load((const u32x4*)src + i);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rb;
         i += (int64_t)gridDim.x * blockDim.x)
      dst[i] = src[i];
  }
}

}  // namespace

extern "C" int dz_replay_sample_uniform(const dz_field_t* fields, int num_fields,
                                        const int64_t* pos_host, int batch, int64_t t,
                                        int64_t size, int64_t capacity, int64_t* ids_out,
                                        dz_stream_t stream) {
  DZ_REQUIRE(fields && pos_host && num_fields > 0 && num_fields <= DZ_MAX_FIELDS);
  DZ_REQUIRE(batch > 0 && batch <= kMaxHostPos && capacity > 0 && size > 0 &&
             size <= capacity && t >= size);
  GatherArgs a;
  a.num_fields = num_fields;
  int64_t max_rb = 0;
  for (int i = 0; i < num_fields; ++i) {
    DZ_REQUIRE(fields[i].src && fields[i].dst && fields[i].row_bytes > 0);
    a.f[i] = fields[i];
    if (fields[i].row_bytes > max_rb) max_rb = fields[i].row_bytes;
  }
  HostPositions hp;
  for (int i = 0; i < kMaxHostPos; ++i) {
    const int64_t p = pos_host[i < batch ? i : 0];
    DZ_REQUIRE(p >= 0 && p < size);
    hp.pos[i] = p;
  }
  int64_t chunks = ((max_rb >> 4) + 255) / 256;
  if (chunks < 1) chunks = 1;
  if (chunks > 64) chunks = 64;
  dz_prof_pair(1, 0, dz_s(stream));
  hipLaunchKernelGGL(gather_uniform_kernel, dim3((unsigned)chunks, (unsigned)batch,
                                                 (unsigned)num_fields),
                     dim3(256), 0, dz_s(stream), a, hp, t, capacity, ids_out);
  DZ_LAUNCH_CHECK();
  dz_prof_pair(1, 1, dz_s(stream));
  return DZ_OK;
}

extern "C" int dz_replay_gather(const dz_field_t* fields, int num_fields,
                                const int64_t* ids, int batch, int64_t capacity,
                                dz_stream_t stream) {
  DZ_REQUIRE(fields && ids && num_fields > 0 && num_fields <= DZ_MAX_FIELDS);
  DZ_REQUIRE(batch > 0 && capacity > 0);
  GatherArgs a;
  a.num_fields = num_fields;
  int64_t max_rb = 0;
  for (int i = 0; i < num_fields; ++i) {
    DZ_REQUIRE(fields[i].src && fields[i].dst && fields[i].row_bytes > 0);
    a.f[i] = fields[i];
    if (fields[i].row_bytes > max_rb) max_rb = fields[i].row_bytes;
  }
  int64_t chunks = ((max_rb >> 4) + 255) / 256;
  if (chunks < 1) chunks = 1;
  if (chunks > 64) chunks = 64;
  dim3 grid((unsigned)chunks, (unsigned)batch, (unsigned)num_fields);
  dz_prof_pair(1, 0, dz_s(stream));
  hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(256), 0, dz_s(stream), a,
                     ids, capacity);
  DZ_LAUNCH_CHECK();
  dz_prof_pair(1, 1, dz_s(stream));
  return DZ_OK;
}

extern "C" int dz_uniform_pos_to_id(const int64_t* pos, int batch, int64_t t,
                                    int64_t size, int64_t capacity,
                                    int64_t* ids_out, dz_stream_t stream) {
  DZ_REQUIRE(pos && ids_out && batch > 0 && capacity > 0 && size > 0 &&
             size <= capacity && t >= size);
  hipLaunchKernelGGL(pos_to_id_kernel, dim3((batch + 255) / 256), dim3(256), 0,
                     dz_s(stream), pos, batch, t, size, capacity, ids_out);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
