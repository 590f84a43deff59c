// ea_gn_stats_loops.h -- CODE FRAGMENT, included inside ea_gn_stats_kernel's body by the reproducer builds only
// (tools/build_gn_repro.sh: -DEA_GN_STATS_LOOP=n; the product is built without the macro and never sees this file).
// The round-3 form of the statistics loop (per-THREAD bound `px < p_end`): every trip ends with the sum-of-squares updates
// (v_pk_fma_f32 x3, v_pk_add_f32) directly followed by `s_andn2_b64 exec` -- the form whose lanes 48..63 occasionally lost
// those last updates beside another stream's generic-kernel launches (profiles/HISTORY.md 8f-1, profiles/r04_pipelined_race.jsonl).
//   EA_GN_STATS_LOOP == 1   the round-3 loop as it shipped
//   EA_GN_STATS_LOOP == 2   the same with wait states pinned between the last updates and the EXEC update
//   EA_GN_STATS_LOOP == 3   the same source as 1; the build adds -fno-slp-vectorize (no packed fp32 instructions)
//   EA_GN_STATS_LOOP == 4   per-thread bound, but the LAST instructions of a trip are the plain-sum updates (s after q)
//   EA_GN_STATS_LOOP == 5   the round-3 loop with its packed updates, wait states behind them (operand-free asm at the loop end)
  for (int px = p_begin + pr; px < p_end; px += 4 * p.R) {
    f16x8 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = px + u * p.R;
      x[u] = in.load(pix0 + (pu < p_end ? pu : p_end - 1));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float m = (px + u * p.R < p_end) ? 1.0f : 0.0f;
#if EA_GN_STATS_LOOP == 4
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { f[j] = (float)x[u][j] * m; q[j] += f[j] * f[j]; }
#if !defined(EA_EMU)
      asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]));
#endif
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j];
#else
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)x[u][j] * m;
        s[j] += f;
        q[j] += f * f;
      }
#endif
    }
#if EA_GN_STATS_LOOP == 2 && !defined(EA_EMU)
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]),
                 "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]));
#endif
#if EA_GN_STATS_LOOP == 5 && !defined(EA_EMU)
    {
      f32x2 q01 = {q[0], q[1]}, q23 = {q[2], q[3]}, q45 = {q[4], q[5]}, q67 = {q[6], q[7]};
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(q01), "+v"(q23), "+v"(q45), "+v"(q67));
      q[0] = q01[0]; q[1] = q01[1]; q[2] = q23[0]; q[3] = q23[1]; q[4] = q45[0]; q[5] = q45[1]; q[6] = q67[0]; q[7] = q67[1];
    }
#endif
  }
