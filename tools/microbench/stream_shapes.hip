// Micro-benchmark: how fast can one kernel stream d+1 = 9 (id, weight) planes of N entries, by access shape?
// Build: hipcc -O3 --offload-arch=gfx950 stream_shapes.hip -o stream_shapes ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int D1 = 9;

// A: thread per point, 4-byte loads from 2*D1 planes
__global__ __launch_bounds__(256) void k_scalar(const int *__restrict__ id, const float *__restrict__ w, int n, float *__restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int v[D1]; float ww[D1];
#pragma unroll
    for (int r = 0; r < D1; ++r) { v[r] = id[(size_t)r * n + p]; ww[r] = w[(size_t)r * n + p]; }
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < D1; ++r) acc += ww[r] * (float)(v[r] & 7);
    out[p] = acc;
}
// B: thread per 4 points, 16-byte loads
__global__ __launch_bounds__(256) void k_vec4(const int *__restrict__ id, const float *__restrict__ w, int n, float *__restrict__ out)
{
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= n) return;
    int4 v[D1]; float4 ww[D1];
#pragma unroll
    for (int r = 0; r < D1; ++r) { v[r] = *(const int4 *)(id + (size_t)r * n + p); ww[r] = *(const float4 *)(w + (size_t)r * n + p); }
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < D1; ++r) { acc.x += ww[r].x * (float)(v[r].x & 7); acc.y += ww[r].y * (float)(v[r].y & 7); acc.z += ww[r].z * (float)(v[r].z & 7); acc.w += ww[r].w * (float)(v[r].w & 7); }
    *(float4 *)(out + p) = acc;
}
// C: AoS: per point a record of D1 ids then D1 weights (72 B), thread per point
__global__ __launch_bounds__(256) void k_aos(const int *__restrict__ rec, int n, float *__restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int *q = rec + (size_t)p * 2 * D1;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < D1; ++r) acc += __int_as_float(q[D1 + r]) * (float)(q[r] & 7);
    out[p] = acc;
}
// D: flat stream copy-like: each thread 16-byte loads from a single 72 MB array (the ceiling)
__global__ __launch_bounds__(256) void k_flat(const float4 *__restrict__ a, size_t n4, float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (size_t k = i; k < n4; k += (size_t)gridDim.x * 256) { float4 v = a[k]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
// E: 2-byte ids + 4-byte weights, thread per point (slice_block shape, no LDS)
__global__ __launch_bounds__(256) void k_u16(const uint16_t *__restrict__ id, const float *__restrict__ w, int n, float *__restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int v[D1]; float ww[D1];
#pragma unroll
    for (int r = 0; r < D1; ++r) { v[r] = id[(size_t)r * n + p]; ww[r] = w[(size_t)r * n + p]; }
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < D1; ++r) acc += ww[r] * (float)(v[r] & 7);
    out[p] = acc;
}
// F: thread per 2 points, 8-byte loads
__global__ __launch_bounds__(256) void k_vec2(const int *__restrict__ id, const float *__restrict__ w, int n, float *__restrict__ out)
{
    const int p = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (p >= n) return;
    int2 v[D1]; float2 ww[D1];
#pragma unroll
    for (int r = 0; r < D1; ++r) { v[r] = *(const int2 *)(id + (size_t)r * n + p); ww[r] = *(const float2 *)(w + (size_t)r * n + p); }
    float2 acc = make_float2(0, 0);
#pragma unroll
    for (int r = 0; r < D1; ++r) { acc.x += ww[r].x * (float)(v[r].x & 7); acc.y += ww[r].y * (float)(v[r].y & 7); }
    *(float2 *)(out + p) = acc;
}
// G: scalar loads + a random 4-byte gather per corner from a table of `m` floats (the old slice shape)
__global__ __launch_bounds__(256) void k_gather(const int *__restrict__ id, const float *__restrict__ w, int n, const float *__restrict__ table, float *__restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int v[D1]; float ww[D1], g[D1];
#pragma unroll
    for (int r = 0; r < D1; ++r) { v[r] = id[(size_t)r * n + p]; ww[r] = w[(size_t)r * n + p]; }
#pragma unroll
    for (int r = 0; r < D1; ++r) g[r] = table[v[r]];
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < D1; ++r) acc += ww[r] * g[r];
    out[p] = acc;
}

template <class F> float timeit(F f, int reps = 30)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main()
{
    const int n = 1000000, m = 400000;
    int *id; float *w, *out, *table; uint16_t *id16;
    CK(hipMalloc(&id, (size_t)2 * D1 * n * 4)); CK(hipMemset(id, 0, (size_t)2 * D1 * n * 4));
    CK(hipMalloc(&w, (size_t)D1 * n * 4)); CK(hipMalloc(&out, (size_t)n * 4 + 64));
    CK(hipMalloc(&table, (size_t)m * 4)); CK(hipMalloc(&id16, (size_t)D1 * n * 2));
    std::vector<int> h((size_t)D1 * n);
    // ids with the locality of a lattice-ordered cloud: a slowly moving base + small offsets
    for (size_t i = 0; i < h.size(); ++i) { size_t p = i % n; h[i] = (int)(((p * (size_t)m) / n + (i * 2654435761u) % 3000) % m); }
    CK(hipMemcpy(id, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(w, 0, (size_t)D1 * n * 4)); CK(hipMemset(table, 0, (size_t)m * 4)); CK(hipMemset(id16, 0, (size_t)D1 * n * 2));
    const double MB = 2.0 * D1 * n * 4 / 1e6;
    float t;
    t = timeit([&] { k_scalar<<<(n + 255) / 256, 256>>>(id, w, n, out); }); printf("A scalar 4B planes   : %7.2f us  %6.0f GB/s\n", t, MB / t * 1e3);
    t = timeit([&] { k_vec2<<<(n / 2 + 255) / 256, 256>>>(id, w, n, out); }); printf("F vec2 8B planes     : %7.2f us  %6.0f GB/s\n", t, MB / t * 1e3);
    t = timeit([&] { k_vec4<<<(n / 4 + 255) / 256, 256>>>(id, w, n, out); }); printf("B vec4 16B planes    : %7.2f us  %6.0f GB/s\n", t, MB / t * 1e3);
    t = timeit([&] { k_aos<<<(n + 255) / 256, 256>>>(id, n, out); }); printf("C AoS 72B records    : %7.2f us  %6.0f GB/s\n", t, MB / t * 1e3);
    t = timeit([&] { k_flat<<<2048, 256>>>((const float4 *)id, (size_t)D1 * n * 2 / 4, out); }); printf("D flat 16B one array : %7.2f us  %6.0f GB/s\n", t, MB / t * 1e3);
    t = timeit([&] { k_u16<<<(n + 255) / 256, 256>>>(id16, w, n, out); }); printf("E u16+f32 planes     : %7.2f us  %6.0f GB/s (54 MB)\n", t, 54.0 / t * 1e3);
    t = timeit([&] { k_gather<<<(n + 255) / 256, 256>>>(id, w, n, table, out); }); printf("G scalar + 9 gathers : %7.2f us  %6.0f GB/s\n", t, MB / t * 1e3);
    return 0;
}
