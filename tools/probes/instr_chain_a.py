"""s_memtime stamps inside chain_a_kernel (csrc/chain.hip + chain_dev.h): writes instrumented COPIES of both files (string
anchors of the commit this file belongs to); the kernel stores per-phase cycle counts into the first 12 floats of the attn row
of each tile's first row (call it without a processing order).

    cd mvgformer_amd/csrc && python ../../tools/probes/instr_chain_a.py
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -c chain_exp.hip -o /tmp/chain_e.o
    hipcc --offload-arch=gfx950 -shared -o ../_exp_CA.so api.o msda.o geom.o gemm.o /tmp/chain_e.o wreg_gemm.o msda_bwd.o
    (GPU box)  cp mvgformer_amd/_exp_CA.so mvgformer_amd/libmvgformer_hip.so; python tools/probes/time_chain_a.py"""
s = open("chain.hip").read()
d = open("chain_dev.h").read()


def rep(txt, a, b):
    assert a in txt, a[:60]
    return txt.replace(a, b, 1)


s = rep(s, '#include "chain_dev.h"', '#include "chain_dev_exp.h"')
s = rep(s, '''  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * RM;

  // Tile row i works on global row order[r0 + i]''', '''  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * RM;
  long ts[12];
  ts[0] = __builtin_amdgcn_s_memtime();

  // Tile row i works on global row order[r0 + i]''')
s = rep(s, '''  chain_a_body<RM, NT, JN, true>(act, rid, w2s, inside, Wp, bp, W0, b0, W1, b1, b2, attn, o, pf1, keepf);
}''', '''  ts[1] = __builtin_amdgcn_s_memtime();
  chain_a_body<RM, NT, JN, true>(ts, act, rid, w2s, inside, Wp, bp, W0, b0, W1, b1, b2, attn, o, pf1, keepf);
  ts[10] = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (tid == 0 && rid[0] >= 0) {
    float* dbg = reinterpret_cast<float*>(attn + (long)rid[0] * 256);
    dbg[0] = -12345.f;
    for (int i = 1; i <= 10; ++i) dbg[i] = (float)(ts[i] - ts[i - 1]);
  }
}''')
d = rep(d, '''__device__ __forceinline__ void chain_a_body(char* __restrict__ act,''', '''__device__ __forceinline__ void chain_a_body(long (&ts)[12], char* __restrict__ act,''')
d = rep(d, '''  stage_gemm<MT, 16, 4, JN, PRE1>(act, Wp, acc, tid, true, rot, 16 * 1024, pf1);
  load_bias<JN>(bp, bvr, tid, MT * 32);''', '''  stage_gemm<MT, 16, 4, JN, PRE1>(act, Wp, acc, tid, true, rot, 16 * 1024, pf1);
  ts[2] = __builtin_amdgcn_s_memtime();
  load_bias<JN>(bp, bvr, tid, MT * 32);''')
d = rep(d, '''  write_act_pre<MT, JN>(act, acc, bvr, false, keep, tid);
  __syncthreads();''', '''  write_act_pre<MT, JN>(act, acc, bvr, false, keep, tid);
  __syncthreads();
  ts[3] = __builtin_amdgcn_s_memtime();''')
d = rep(d, '''  // pose_embed MLP layers 0, 1 (ReLU)
  stage_gemm<MT, 16, 4, JN, true>(act, W0, acc, tid, true, rot + 5, 16 * 1024, pf);''', '''  ts[4] = __builtin_amdgcn_s_memtime();
  // pose_embed MLP layers 0, 1 (ReLU)
  stage_gemm<MT, 16, 4, JN, true>(act, W0, acc, tid, true, rot + 5, 16 * 1024, pf);
  ts[5] = __builtin_amdgcn_s_memtime();''')
d = rep(d, '''  write_act_pre<MT, JN>(act, acc, bvr, true, all, tid);
  __syncthreads();
  stage_gemm<MT, 16, 4, JN, true>(act, W1, acc, tid, true, rot + 10, 16 * 1024, pf);
  __syncthreads();''', '''  write_act_pre<MT, JN>(act, acc, bvr, true, all, tid);
  __syncthreads();
  ts[6] = __builtin_amdgcn_s_memtime();
  stage_gemm<MT, 16, 4, JN, true>(act, W1, acc, tid, true, rot + 10, 16 * 1024, pf);
  __syncthreads();
  ts[7] = __builtin_amdgcn_s_memtime();''')
d = rep(d, '''  write_act<MT, JN>(act, acc, b1, true, all, tid);
  __syncthreads();''', '''  write_act<MT, JN>(act, acc, b1, true, all, tid);
  __syncthreads();
  ts[8] = __builtin_amdgcn_s_memtime();''')
d = rep(d, '''  if (part == 0 && rid[row] >= 0) {
    float* og = o + (long)rid[row] * 3;''', '''  ts[9] = __builtin_amdgcn_s_memtime();
  if (part == 0 && rid[row] >= 0) {
    float* og = o + (long)rid[row] * 3;''')
open("chain_exp.hip", "w").write(s)
open("chain_dev_exp.h", "w").write(d)
