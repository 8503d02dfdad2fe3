// Hardware-semantics probes for gfx950 (run once on the GPU box; results recorded in DESIGN.md).
//  1. v_mfma_f32_32x32x16_bf16 / 16x16x32 operand + accumulator lane layouts
//  2. ds_read_b64_tr_b16 gather pattern
//  3. global_load_lds_dwordx4 destination pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static float bf2f(unsigned short b) { unsigned u = ((unsigned)b) << 16; float f; memcpy(&f, &u, 4); return f; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// A: [32][16] row-major bf16, B: [16][32] row-major bf16 (k rows), D: [32][32] fp32
__global__ void mfma32(const unsigned short* A, const unsigned short* B, float* D) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j];
    b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)];
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    D[row * 32 + col] = c[r];
  }
}
// A: [16][32], B: [32][16], D [16][16]
__global__ void mfma16(const unsigned short* A, const unsigned short* B, float* D) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
  }
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

__global__ void trread(unsigned short* out /*[64][4]*/, int row_stride_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  // natural per-lane address: 16-lane group g reads block g; lane t in group: row t>>2, 8-byte piece t&3
  const int g = l >> 4, t = l & 15;
  // LDS byte address = low 32 bits of the address_space(3) pointer; derived from `lds` so the
  // fill above stays live
  unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
  addr += (unsigned)(g * 4 * row_stride_bytes + (t >> 2) * row_stride_bytes + (t & 3) * 8);
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}

__global__ void glds(const unsigned* src /*[64*4 + extra]*/, unsigned* out /*[512]*/) {
  __shared__ __attribute__((aligned(16))) unsigned lds[512];
  const int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  // each lane supplies its own 16-byte source: lane l reads src[(63-l)*4 ..]
  const unsigned* g = src + (63 - l) * 4;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(lds + 64), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 512; i += 64) out[i] = lds[i];
}

int main() {
  // ---- MFMA 32x32x16
  {
    std::vector<unsigned short> A(32 * 16), B(16 * 32);
    std::vector<float> Af(32 * 16), Bf(16 * 32), D(32 * 32), R(32 * 32, 0.f);
    srand(1);
    for (int i = 0; i < 32 * 16; ++i) { A[i] = f2bf((rand() % 17 - 8) / 4.f); Af[i] = bf2f(A[i]); }
    for (int i = 0; i < 16 * 32; ++i) { B[i] = f2bf((rand() % 13 - 6) / 2.f); Bf[i] = bf2f(B[i]); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
    unsigned short *dA, *dB; float* dD;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    mfma32<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("PROBE mfma_f32_32x32x16_bf16 assumed layout: max_err=%g %s\n", err, err < 1e-3 ? "PASS" : "FAIL");
  }
  // ---- MFMA 16x16x32
  {
    std::vector<unsigned short> A(16 * 32), B(32 * 16);
    std::vector<float> Af(16 * 32), Bf(32 * 16), D(256), R(256, 0.f);
    srand(2);
    for (int i = 0; i < 512; ++i) { A[i] = f2bf((rand() % 17 - 8) / 4.f); Af[i] = bf2f(A[i]); }
    for (int i = 0; i < 512; ++i) { B[i] = f2bf((rand() % 13 - 6) / 2.f); Bf[i] = bf2f(B[i]); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += Af[i * 32 + k] * Bf[k * 16 + j];
    unsigned short *dA, *dB; float* dD;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    mfma16<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("PROBE mfma_f32_16x16x32_bf16 assumed layout: max_err=%g %s\n", err, err < 1e-3 ? "PASS" : "FAIL");
  }
  // ---- ds_read_b64_tr_b16
  for (int stride : {32, 64, 128}) {
    unsigned short* d; CK(hipMalloc(&d, 64 * 4 * 2));
    trread<<<1, 64>>>(d, stride);
    std::vector<unsigned short> h(256);
    CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
    printf("PROBE ds_read_b64_tr_b16 row_stride=%dB (lds element index per lane, 4 values):\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d: %4d %4d %4d %4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
      if (l % 2 == 1) printf("\n");
    }
    // check the hypothesis: lane t of group g gets column t of the [4][16] block: rows 0..3
    bool ok = true;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int g = l >> 4, t = l & 15;
      int expect = (g * 4 * stride + j * stride) / 2 + t;  // column t of the group's 4x16 block
      if (h[l * 4 + j] != expect) ok = false;
    }
    printf("PROBE tr_b16 hypothesis(column t of 4x16 block): %s\n", ok ? "PASS" : "FAIL");
  }
  // ---- global_load_lds
  {
    std::vector<unsigned> src(64 * 4);
    for (int i = 0; i < 256; ++i) src[i] = i;
    unsigned *ds, *dout; CK(hipMalloc(&ds, 1024)); CK(hipMalloc(&dout, 2048));
    CK(hipMemcpy(ds, src.data(), 1024, hipMemcpyHostToDevice));
    glds<<<1, 64>>>(ds, dout);
    std::vector<unsigned> out(512);
    CK(hipMemcpy(out.data(), dout, 2048, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (out[64 + l * 4 + j] != (unsigned)((63 - l) * 4 + j)) ok = false;
    for (int i = 0; i < 64; ++i) if (out[i] != 0xdeadbeefu) ok = false;
    for (int i = 320; i < 512; ++i) if (out[i] != 0xdeadbeefu) ok = false;
    printf("PROBE global_load_lds_dwordx4 dest = base + lane*16, per-lane source: %s (first words: %u %u %u %u %u)\n",
           ok ? "PASS" : "FAIL", out[64], out[65], out[68], out[72], out[316]);
  }
  return 0;
}
