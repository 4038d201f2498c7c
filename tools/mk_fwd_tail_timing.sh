#!/bin/bash
# dev helper: the timing build of tools/mk_fwd_timing.sh plus stage stamps inside the fused tail's shading rounds
# (read with tools/fwd_tail_stages.py from a dump of tools/fwd_phase_times.py).  Output abx/libf_tim3.so
NAME=${NAME:-f_tim3} bash "$(dirname "$0")/mk_fwd_timing.sh" "$@" \
 '      int il[U], ik[U], ifid[U];@@      bool act[U];=>      unsigned long long t_r = wt ? clock64() : 0ull;@@      int il[U], ik[U], ifid[U];@@      bool act[U];' \
 '        ifid[u] = key_face(key);@@      }=>        ifid[u] = key_face(key);@@      }@@      if (wt) { const unsigned long long t_n = clock64(); d_lock += t_n - t_r; t_r = t_n; }' \
 '#pragma unroll@@      for (int u = 0; u < U; ++u) sample_rgb(=>      if (wt) { const unsigned long long t_n = clock64(); d_comp += t_n - t_r; t_r = t_n; }@@#pragma unroll@@      for (int u = 0; u < U; ++u) sample_rgb(' \
 '        s_res[64 * u + lane] = make_float4(va[u], vr[u], vg[u], vb[u]);@@      }@@      wave_lds_handoff();=>        s_res[64 * u + lane] = make_float4(va[u], vr[u], vg[u], vb[u]);@@      }@@      wave_lds_handoff();@@      if (wt) { const unsigned long long t_n = clock64(); n_lock += t_n - t_r; t_r = t_n; }' \
 '      // the layer rows for the backward, written by the lanes that computed them=>      if (wt) { const unsigned long long t_n = clock64(); n_comp += t_n - t_r; t_r = t_n; }@@      // the layer rows for the backward, written by the lanes that computed them' \
 '          row[1] = make_int4(__float_as_int(vr[u]), __float_as_int(vg[u]), __float_as_int(vb[u]), 0);@@        }@@    };=>          row[1] = make_int4(__float_as_int(vr[u]), __float_as_int(vg[u]), __float_as_int(vb[u]), 0);@@        }@@      if (wt) { const unsigned long long t_n = clock64(); d_scan += t_n - t_r; }@@    };' \
 '    int base = 0;@@    for (; base + 64 < total; base += 128)=>    d_scan = 0;@@    int base = 0;@@    for (; base + 64 < total; base += 128)' \
 '    if (pix_live) {@@      const int64_t hw = (int64_t)H * W;@@      float* img = P.image=>    if (wt && lane == 0) { wt[6] = d_lock; wt[7] = d_comp; wt[9] = n_lock; wt[12] = n_comp; wt[13] = d_scan; }@@    if (pix_live) {@@      const int64_t hw = (int64_t)H * W;@@      float* img = P.image'
