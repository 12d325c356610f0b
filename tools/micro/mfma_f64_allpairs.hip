// Microbenchmark (diagnostic): the all-pairs blocks of the tick -- per robot T[12 x 13] = Yrows[12 x 6] . [B | ab0][6 x 13]
// (the J'WJ-type contraction north_star assigns to MFMA; reference: controllers/mptc_controller.py:51-55) -- done
//   (i)  as the product kernel does it: 16 lanes per robot, one COLUMN of [B | ab0] per lane, Y rows broadcast from
//        their owner lanes by the fused v_fmac_f64_dpp row_newbcast (72 fused ops per lane, three rows per asm block);
//   (ii) on the matrix core: v_mfma_f64_4x4x4f64 (four independent 4x4x4 products per instruction, one per robot of the
//        wavefront), 3 x 4 x 2 = 24 tiles per robot, operands and results moved between the kernel's column-per-lane
//        register layout and the instruction's tile layout through LDS, INCLUDING those moves -- the contraction is only
//        useful to the tick in the layout the QR consumes.  The instruction's lane layout is PROBED at start-up (one-hot
//        operands): a block's operands sit on lanes {4b..4b+3} of each of the four 16-lane rows (A: i = lane % 4,
//        k = lane / 16; B: j = lane % 4, k = lane / 16; D: j = lane % 4, i = lane / 16), i.e. a block does NOT coincide with
//        a DPP row = a robot, so every operand crosses rows: only LDS (or ds_bpermute) can feed it.
// Both start from and end in the same registers (Yrow[6], bcol[6] -> T[12] per lane), are checked against each other and a
// scalar reference, and are timed as REPS back-to-back evaluations per wavefront (4 robots) with one wavefront per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_micro tools/micro/mfma_f64_allpairs.hip && /tmp/mfma_micro
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ constexpr int hex_lane(int k) { return k < 12 ? 4 * (k / 3) + (k % 3) : 3; }   // column 12 (rhs) on lane 3

// ---- lane layout of v_mfma_f64_4x4x4f64, discovered by probing: D[la][lb][lane] for one-hot A (lane la) and B (lane lb)
__global__ void probe_kernel(double* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(la * 64 + lb) * 64 + lane] = d;
    }
}

struct Maps { int bA[64], iA[64], kA[64], bB[64], kB[64], jB[64], bD[64], iD[64], jD[64]; };   // block = which of the 4 independent products
__constant__ Maps c_maps;

// (i) DPP path ------------------------------------------------------------------------------------------------------
#define WBC_FP(A, X, Y, L) "v_fmac_f64_dpp %" #A ", %" #X ", %" #Y " row_newbcast:%[" #L "] row_mask:0xf bank_mask:0xf\n\t"
template <int L0, int L1, int L2> __device__ __forceinline__ void rows3_bc(double& d0, double& d1, double& d2, const double* x, const double* y) {
  asm volatile(WBC_FP(0, 3, 9, l0) WBC_FP(1, 3, 9, l1) WBC_FP(2, 3, 9, l2) WBC_FP(0, 4, 10, l0) WBC_FP(1, 4, 10, l1) WBC_FP(2, 4, 10, l2)
               WBC_FP(0, 5, 11, l0) WBC_FP(1, 5, 11, l1) WBC_FP(2, 5, 11, l2) WBC_FP(0, 6, 12, l0) WBC_FP(1, 6, 12, l1) WBC_FP(2, 6, 12, l2)
               WBC_FP(0, 7, 13, l0) WBC_FP(1, 7, 13, l1) WBC_FP(2, 7, 13, l2) WBC_FP(0, 8, 14, l0) WBC_FP(1, 8, 14, l1) WBC_FP(2, 8, 14, l2)
               : "+v"(d0), "+v"(d1), "+v"(d2)
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]),
                 [l0] "n"(L0), [l1] "n"(L1), [l2] "n"(L2));
}
__device__ __forceinline__ void allpairs_dpp(const double* yrow, const double* bcol, double* T) {
  double y[6] = {yrow[0], yrow[1], yrow[2], yrow[3], yrow[4], yrow[5]};
  asm volatile("s_nop 4" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]));   // hazard fence (DPP source)
  for (int r = 0; r < 12; r++) T[r] = 0.0;
  rows3_bc<hex_lane(0), hex_lane(1), hex_lane(2)>(T[0], T[1], T[2], y, bcol);
  rows3_bc<hex_lane(3), hex_lane(4), hex_lane(5)>(T[3], T[4], T[5], y, bcol);
  rows3_bc<hex_lane(6), hex_lane(7), hex_lane(8)>(T[6], T[7], T[8], y, bcol);
  rows3_bc<hex_lane(9), hex_lane(10), hex_lane(11)>(T[9], T[10], T[11], y, bcol);
}

// (ii) MFMA path ----------------------------------------------------------------------------------------------------
// LDS per robot: Y[12][8] (K padded to 8) and Bm[8][16] (13 columns padded to 16), then D[12][16].
__device__ __forceinline__ void allpairs_mfma(const double* yrow, const double* bcol, double* T, double (*ldsY)[12 * 8], double (*ldsB)[8 * 16],
                                               double (*ldsD)[12 * 16], int lane, int rob, int myrow, int mycol) {
  // out: every lane that owns a row / a column writes its 6 values (the padding was zeroed once, outside the timed loop)
  if (myrow >= 0)
    for (int k = 0; k < 6; k++) ldsY[rob][myrow * 8 + k] = yrow[k];
  if (mycol >= 0)
    for (int k = 0; k < 6; k++) ldsB[rob][k * 16 + mycol] = bcol[k];
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0); one wavefront, no barrier needed
  // the instruction's four independent products (blocks) are served one robot each: lane l supplies block bA/bB's operands
  const int bA = c_maps.bA[lane], iA = c_maps.iA[lane], kA = c_maps.kA[lane];
  const int bB = c_maps.bB[lane], kB = c_maps.kB[lane], jB = c_maps.jB[lane];
  const int bD = c_maps.bD[lane], iD = c_maps.iD[lane], jD = c_maps.jD[lane];
  double a[3][2], b[4][2];
  for (int rt = 0; rt < 3; rt++)
    for (int kt = 0; kt < 2; kt++) a[rt][kt] = ldsY[bA][(4 * rt + iA) * 8 + 4 * kt + kA];
  for (int ct = 0; ct < 4; ct++)
    for (int kt = 0; kt < 2; kt++) b[ct][kt] = ldsB[bB][(4 * kt + kB) * 16 + 4 * ct + jB];
  for (int rt = 0; rt < 3; rt++)
    for (int ct = 0; ct < 4; ct++) {
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a[rt][0], b[ct][0], 0.0, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f64_4x4x4f64(a[rt][1], b[ct][1], d, 0, 0, 0);
      ldsD[bD][(4 * rt + iD) * 16 + 4 * ct + jD] = d;
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  const int c = (mycol >= 0) ? mycol : 15;
  for (int r = 0; r < 12; r++) T[r] = ldsD[rob][r * 16 + c];
}

template <int MODE> __global__ void __launch_bounds__(64)
bench_kernel(const double* yin, const double* bin, double* out, unsigned long long* cyc, int reps) {
  __shared__ double ldsY[4][12 * 8], ldsB[4][8 * 16], ldsD[4][12 * 16];
  const int lane = threadIdx.x, lane16 = lane & 15, rob = lane >> 4;
  // row r of Y lives on lane hex_lane(r) (r < 12); column c of [B | ab0] on lane hex_lane(c) (c < 13)
  int myrow = -1, mycol = -1;
  for (int r = 0; r < 12; r++) if (hex_lane(r) == lane16) myrow = r;
  for (int c = 0; c < 13; c++) if (hex_lane(c) == lane16) mycol = c;
  double yrow[6], bcol[6], T[12], acc = 0.0;
  const size_t base = ((size_t)blockIdx.x * 64 + lane) * 6;
  for (int k = 0; k < 6; k++) { yrow[k] = yin[base + k]; bcol[k] = bin[base + k]; }
  for (int i = lane16; i < 12 * 8; i += 16) ldsY[rob][i] = 0.0;
  for (int i = lane16; i < 8 * 16; i += 16) ldsB[rob][i] = 0.0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < reps; it++) {
    if (MODE == 0) allpairs_dpp(yrow, bcol, T);
    else allpairs_mfma(yrow, bcol, T, ldsY, ldsB, ldsD, lane, rob, myrow, mycol);
    double s = 0.0;
    for (int r = 0; r < 12; r++) s += T[r];
    acc += s;
    yrow[it % 6] += 1e-9 * s;      // loop-carried: no hoisting, no dead-code elimination
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < 12; r++) out[((size_t)blockIdx.x * 64 + lane) * 13 + r] = T[r];
  out[((size_t)blockIdx.x * 64 + lane) * 13 + 12] = acc;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024, reps = argc > 2 ? atoi(argv[2]) : 2000;
  // ---- probe the instruction's layout
  double* dprobe; CHECK(hipMalloc(&dprobe, 64 * 64 * 64 * 8));
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dprobe);
  std::vector<double> P(64 * 64 * 64);
  CHECK(hipMemcpy(P.data(), dprobe, P.size() * 8, hipMemcpyDeviceToHost));
  auto D = [&](int la, int lb, int l) { return P[(la * 64 + lb) * 64 + l]; };
  Maps M;
  // Every (A lane, B lane) pair lights at most ONE D lane (same block, same k).  Blocks = connected components of the D lanes
  // under "lit by the same A lane" / "lit by the same B lane"; rows / columns of a block = the D lanes one A / B lane lights.
  int par[64]; for (int l = 0; l < 64; l++) par[l] = l;
  auto find = [&](int x) { while (par[x] != x) x = par[x] = par[par[x]]; return x; };
  std::vector<int> litA[64], litB[64];
  for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) for (int l = 0; l < 64; l++) if (D(la, lb, l) != 0.0) { litA[la].push_back(l); litB[lb].push_back(l); }
  int ok = 1;
  for (int x = 0; x < 64; x++) {
    if (litA[x].size() != 4 || litB[x].size() != 4) ok = 0;
    for (int l : litA[x]) par[find(l)] = find(litA[x][0]);
    for (int l : litB[x]) par[find(l)] = find(litB[x][0]);
  }
  int blkid[64], nblk = 0; for (int l = 0; l < 64; l++) blkid[l] = -1;
  for (int l = 0; l < 64; l++) { int r = find(l); if (blkid[r] < 0) blkid[r] = nblk++; M.bD[l] = blkid[r]; }
  int rowof[64], colof[64]; for (int l = 0; l < 64; l++) rowof[l] = colof[l] = -1;
  int nrow[4] = {0, 0, 0, 0}, ncol[4] = {0, 0, 0, 0};
  for (int la = 0; la < 64 && ok; la++) {
    const int b = M.bD[litA[la][0]];
    int g = rowof[litA[la][0]]; if (g < 0) g = nrow[b]++;
    for (int l : litA[la]) rowof[l] = g;
    M.bA[la] = b; M.iA[la] = g;
  }
  for (int lb = 0; lb < 64 && ok; lb++) {
    const int b = M.bD[litB[lb][0]];
    int g = colof[litB[lb][0]]; if (g < 0) g = ncol[b]++;
    for (int l : litB[lb]) colof[l] = g;
    M.bB[lb] = b; M.jB[lb] = g;
  }
  for (int l = 0; l < 64; l++) { M.iD[l] = rowof[l]; M.jD[l] = colof[l]; }
  int cnt[4][4] = {{0}};
  for (int lb = 0; lb < 64 && ok; lb++) M.kB[lb] = cnt[M.bB[lb]][M.jB[lb]]++;
  for (int la = 0; la < 64 && ok; la++) {
    M.kA[la] = -1;
    for (int lb = 0; lb < 64; lb++) { bool hit = false; for (int l = 0; l < 64; l++) hit |= D(la, lb, l) != 0.0; if (hit) M.kA[la] = M.kB[lb]; }
  }
  printf("probe of v_mfma_f64_4x4x4f64: %d independent products (blocks), every operand lane lights 4 result lanes: %s\n", nblk, ok ? "yes" : "NO");
  printf("  lanes of block 0 as A (i,k): "); for (int l = 0; l < 64; l++) if (M.bA[l] == 0) printf("%d:(%d,%d) ", l, M.iA[l], M.kA[l]);
  printf("\n  lanes of block 0 as B (k,j): "); for (int l = 0; l < 64; l++) if (M.bB[l] == 0) printf("%d:(%d,%d) ", l, M.kB[l], M.jB[l]);
  printf("\n  lanes of block 0 as D (i,j): "); for (int l = 0; l < 64; l++) if (M.bD[l] == 0) printf("%d:(%d,%d) ", l, M.iD[l], M.jD[l]);
  printf("\n  => a block is NOT one 16-lane DPP row (= one robot of the kernel's mapping): its operands sit on 4 lanes of each of the four rows\n");
  CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_maps), &M, sizeof M));
  // ---- data
  const size_t n = (size_t)blocks * 64;
  std::vector<double> hy(n * 6), hb(n * 6);
  srand(1);
  for (auto& x : hy) x = rand() / (double)RAND_MAX - 0.5;
  for (auto& x : hb) x = rand() / (double)RAND_MAX - 0.5;
  double *dy, *db, *dout; unsigned long long* dc;
  CHECK(hipMalloc(&dy, n * 6 * 8)); CHECK(hipMalloc(&db, n * 6 * 8)); CHECK(hipMalloc(&dout, n * 13 * 8)); CHECK(hipMalloc(&dc, blocks * 8));
  CHECK(hipMemcpy(dy, hy.data(), n * 6 * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb.data(), n * 6 * 8, hipMemcpyHostToDevice));
  std::vector<double> o0(n * 13), o1(n * 13);
  std::vector<unsigned long long> cy(blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double res[2][2];
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 3; rep++) {   // the last repetition counts
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(bench_kernel<0>, dim3(blocks), dim3(64), 0, 0, dy, db, dout, dc, reps);
      else hipLaunchKernelGGL(bench_kernel<1>, dim3(blocks), dim3(64), 0, 0, dy, db, dout, dc, reps);
      hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    CHECK(hipMemcpy(mode ? o1.data() : o0.data(), dout, n * 13 * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(cy.data(), dc, blocks * 8, hipMemcpyDeviceToHost));
    double mc = 0; for (auto c : cy) mc += (double)c; mc /= blocks;
    res[mode][0] = ms; res[mode][1] = mc / reps;
  }
  // ---- check: one evaluation against a scalar reference (the timed loop perturbs Y, so re-run with reps = 1)
  double worst = 0.0, worst_ref = 0.0;
  for (int mode = 0; mode < 2; mode++) {
    if (mode == 0) hipLaunchKernelGGL(bench_kernel<0>, dim3(blocks), dim3(64), 0, 0, dy, db, dout, dc, 1);
    else hipLaunchKernelGGL(bench_kernel<1>, dim3(blocks), dim3(64), 0, 0, dy, db, dout, dc, 1);
    CHECK(hipMemcpy(mode ? o1.data() : o0.data(), dout, n * 13 * 8, hipMemcpyDeviceToHost));
  }
  for (size_t w = 0; w < (size_t)blocks * 4; w++)          // robot w: lanes 16w .. 16w+15
    for (int c = 0; c < 13; c++)
      for (int r = 0; r < 12; r++) {
        const size_t lr = w * 16 + hex_lane(r), lc = w * 16 + hex_lane(c);
        double ref = 0.0;
        for (int k = 0; k < 6; k++) ref += hy[lr * 6 + k] * hb[lc * 6 + k];
        worst_ref = fmax(worst_ref, fabs(o0[lc * 13 + r] - ref));
        worst = fmax(worst, fabs(o1[lc * 13 + r] - o0[lc * 13 + r]));
      }
  printf("check: |DPP - scalar reference| max %.2e, |MFMA - DPP| max %.2e\n", worst_ref, worst);
  const double flop = 2.0 * 12 * 13 * 6;   // useful flops per robot per evaluation
  for (int mode = 0; mode < 2; mode++)
    printf("%-34s %8.3f ms for %d evaluations x %d wavefronts: %7.1f shader cycles per evaluation per wavefront (4 robots) = %6.1f per robot; %.2f useful TFLOP/s\n",
           mode ? "(ii) v_mfma_f64_4x4x4f64 + LDS moves" : "(i)  v_fmac_f64_dpp row_newbcast", res[mode][0], reps, blocks, res[mode][1], res[mode][1] / 4,
           flop * 4 * blocks * (double)reps / (res[mode][0] * 1e-3) / 1e12);
  printf("ratio (ii) / (i): %.2f in cycles, %.2f in wall time\n", res[1][1] / res[0][1], res[1][0] / res[0][0]);
  return 0;
}
