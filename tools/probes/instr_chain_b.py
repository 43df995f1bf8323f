"""s_memtime stamps inside chain_b_kernel (csrc/chain.hip): writes an instrumented COPY of the source (string anchors of the
commit this file belongs to) whose kernel stores per-phase cycle counts into the first 16 floats of each tile's tgt_out row.

    cd mvgformer_amd/csrc && python ../../tools/probes/instr_chain_b.py chain.hip chain_exp.hip
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -c chain_exp.hip -o /tmp/chain_e.o
    hipcc --offload-arch=gfx950 -shared -o ../_exp_CB.so api.o msda.o geom.o gemm.o /tmp/chain_e.o wreg_gemm.o msda_bwd.o
    (GPU box)  cp mvgformer_amd/_exp_CB.so mvgformer_amd/libmvgformer_hip.so; python tools/probes/time_chain_b.py
Results of round 3: profiles/r03_experiments.txt."""
import sys
src, dst = sys.argv[1], sys.argv[2]
s = open(src).read()
def rep(a, b):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b, 1)
rep('''  const int nrow = min(rpt, rows - r0);
  // All tiles walk the SAME weights;''','''  const int nrow = min(rpt, rows - r0);
  long ts[12];
  ts[0] = __builtin_amdgcn_s_memtime();
  // All tiles walk the SAME weights;''')
rep('''  __syncthreads();

  // ---- u = feature_update_mlp(mean)''','''  __syncthreads();
  ts[1] = __builtin_amdgcn_s_memtime();

  // ---- u = feature_update_mlp(mean)''')
rep('''  acc_to_x<MT, JN>(xb, acc, lnp + 1536, false, tid);
  __syncthreads();
''','''  acc_to_x<MT, JN>(xb, acc, lnp + 1536, false, tid);
  __syncthreads();
  ts[2] = __builtin_amdgcn_s_memtime();
''')
rep('''  __syncthreads();

  // query_pos of the last row phase''','''  __syncthreads();
  ts[3] = __builtin_amdgcn_s_memtime();

  // query_pos of the last row phase''')
rep('''    for (int c = 0; c < 4; ++c) {
      stage_gemm<MT, 16, BRING, JN, true, true>(act, W1''','''    for (int c = 0; c < 4; ++c) {
      if (c == 1) ts[4] = __builtin_amdgcn_s_memtime();
      stage_gemm<MT, 16, BRING, JN, true, true>(act, W1''')
rep('''      write_act_pre<MT, JN>(hbuf, acc, bv1, true, all, tid);                                   // private buffer: no hazard with act
      __syncthreads();''','''      write_act_pre<MT, JN>(hbuf, acc, bv1, true, all, tid);                                   // private buffer: no hazard with act
      __syncthreads();
      if (c == 1) ts[5] = __builtin_amdgcn_s_memtime();''')
rep('''      __syncthreads();                                                               // hbuf free for the next chunk
    }''','''      __syncthreads();                                                               // hbuf free for the next chunk
      if (c == 1) ts[6] = __builtin_amdgcn_s_memtime();
    }''')
rep('''    acc_to_x<MT, JN>(xb, accy, lnp + 1792, true, tid);                                   // x = t1 + Y + b2
    __syncthreads();
  }
''','''    acc_to_x<MT, JN>(xb, accy, lnp + 1792, true, tid);                                   // x = t1 + Y + b2
    __syncthreads();
  }
  ts[7] = __builtin_amdgcn_s_memtime();
''')
rep('''  if (tid < qpt && q0 + tid < nq_total) {''','''  ts[8] = __builtin_amdgcn_s_memtime();
  if (tid < qpt && q0 + tid < nq_total) {''')
rep('''  if (Wn) {
    // ---- xw = (tgt' + query_pos) W_next^T + b_next''','''  ts[9] = __builtin_amdgcn_s_memtime();
  if (Wn) {
    // ---- xw = (tgt' + query_pos) W_next^T + b_next''')
rep('''            *reinterpret_cast<const f32x4*>(xb + row * XP + lane * 16);
  }
}

}  // namespace''','''            *reinterpret_cast<const f32x4*>(xb + row * XP + lane * 16);
  }
  ts[10] = __builtin_amdgcn_s_memtime();
  __syncthreads();
  __threadfence();
  if (tid == 0) {
    float* dbg = tgt_out + (long)r0 * 256;
    for (int i = 1; i <= 10; ++i) dbg[i] = (float)(ts[i] - ts[i - 1]);
    dbg[0] = -12345.f;
  }
}

}  // namespace''')
rep("""    if (has_ffn) {
      const float mean = sum8(sm) * (1.f / 256.f);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        y[i] = y[i] - mean;""","""    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tl[0] = __builtin_amdgcn_s_memtime();
    if (has_ffn) {
      const float mean = sum8(sm) * (1.f / 256.f);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        y[i] = y[i] - mean;""")
rep("""    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = part + 8 * i;
      if (row < nrow) *reinterpret_cast<f32x4*>(tgt_out""","""    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tl[1] = __builtin_amdgcn_s_memtime();
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = part + 8 * i;
      if (row < nrow) *reinterpret_cast<f32x4*>(tgt_out""")
rep("""    a0 = sum8(a0) + bc0;
    a1 = sum8(a1) + bc1;""","""    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tl[2] = __builtin_amdgcn_s_memtime();
    a0 = sum8(a0) + bc0;
    a1 = sum8(a1) + bc1;""")
rep("""  ts[8] = __builtin_amdgcn_s_memtime();
  if (tid < qpt""","""  ts[8] = __builtin_amdgcn_s_memtime();
  tl[4] = ts[8];
  if (tid < qpt""")
rep("""      pr[2 * row + 1] = 1.f / (1.f + expf(-a1));
    }
  }""","""      pr[2 * row + 1] = 1.f / (1.f + expf(-a1));
    }
    tl[3] = __builtin_amdgcn_s_memtime();
  }""")
rep("""  long ts[12];""","""  long ts[12], tl[6];""")
rep("""    dbg[0] = -12345.f;""","""    dbg[0] = -12345.f;
    dbg[11] = (float)(tl[0] - ts[7]); dbg[12] = (float)(tl[1] - tl[0]); dbg[13] = (float)(tl[2] - tl[1]); dbg[14] = (float)(tl[3] - tl[2]); dbg[15] = (float)(tl[4] - tl[3]);""")
open(dst, 'w').write(s)
