import sys
p='/root/repo/robustbnns_amd/csrc/rbnn_conv_x3.hip'
s=open(p).read()
def rep(old,new):
    global s
    assert old in s, old[:60]
    s=s.replace(old,new,1)
rep("""    const int foff = li * 64 + ((lg ^ swz(li)) * 16);
    // image position of output position 16pt + li (tap 0,0); positions past NPOS read (0,0), never stored
    int pbase[NPT], ybase[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        int pos = pt * 16 + li;
        if (pos >= NPOS_) pos = RBNN_CONVX3_OLD_IMG ? 0 : pos - NPOS_;  // idle lanes of the last tile: distinct valid positions (never stored)
        ybase[pt] = pos / O2W_;
        pbase[pt] = ybase[pt] * IPITCH + pos % O2W_;
    }
    const char* const img = imgs + wp * L::IMGB;
""","""    const int foff = li * 64 + ((lg ^ swz(li)) * 16);
#ifndef RBNN_X3FWD_PAIR13
#define RBNN_X3FWD_PAIR13 1
#endif
    constexpr bool PAIR = RBNN_X3FWD_PAIR13 && (NPOS_ % 16 != 0) && !RBNN_CONVX3_OLD_IMG;
    static_assert(!PAIR || (2 * NPOS_ + 15) / 16 == 2 * NPT - 1, "13 = 7 + 6 tiles");
    const char* const img = imgs + wp * L::IMGB;
    constexpr int CPITCH = NPOS_ + 4, EIT = (16 * NP2_ / 4 + 63) / 64;
    auto body = [&](auto NTC) {
    constexpr int NT = decltype(NTC)::value;                              // position tiles of this wave
    const int c0 = PAIR && wp ? 16 * NPT : 0;                              // (PAIR) first combined position of the wave
    int pbase[NT], ybase[NT];
    int ioff_s = 0;
#pragma unroll
    for (int pt = 0; pt < NT; ++pt) {
        int pos = pt * 16 + li;
        if (PAIR) {
            int c = c0 + pos;
            if (c >= 2 * NPOS_) c -= NPOS_;
            const int point = c >= NPOS_ ? 1 : 0;
            pos = c - point * NPOS_;
            if (pt == NT - 1 && NT == NPT) ioff_s = (point - wp) * L::IMGB;
        } else if (pos >= NPOS_) pos = RBNN_CONVX3_OLD_IMG ? 0 : pos - NPOS_;
        ybase[pt] = pos / O2W_;
        pbase[pt] = ybase[pt] * IPITCH + pos % O2W_;
    }
""")
rep("""    // epilogue roles: lane handles the four consecutive pooled cells 4 * (lane + 64 it) .. of a 16-channel tile; their offsets in the wave's tile
    constexpr int CPITCH = NPOS_ + 4, EIT = (16 * NP2_ / 4 + 63) / 64;
    int pbase_e[EIT][4];
""","""    int pbase_e[EIT][4];
""")
rep("""        f32x4 acc[HTW][NPT];
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};""","""        f32x4 acc[HTW][NT];
#pragma unroll
        for (int ht = 0; ht < HTW; ++ht)
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) acc[ht][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};""")
rep("""        constexpr bool PF_ALL = RBNN_X3FWD_BPREFETCH && NPT <= 4;
        auto load_b = [&](int tap, f16x8 (&b0)[NPT], f16x8 (&b1)[NPT], f16x8 (&b2)[NPT], bool lo, bool hi) {
            const int ky = tap / 5, toff = ky * IPITCH + (tap % 5);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int p = pbase[pt] + toff;
                const char* const src = img + p * 64 + ((lg ^ x3_img_swz<G>(p, ybase[pt] + ky)) * 16);""","""        constexpr bool PF_ALL = RBNN_X3FWD_BPREFETCH && NT <= 4;
        auto load_b = [&](int tap, f16x8 (&b0)[NT], f16x8 (&b1)[NT], f16x8 (&b2)[NT], bool lo, bool hi) {
            const int ky = tap / 5, toff = ky * IPITCH + (tap % 5);
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) {
                const int p = pbase[pt] + toff;
                const char* const src = img + ((PAIR && NT == NPT && pt == NT - 1) ? ioff_s : 0) + p * 64 + ((lg ^ x3_img_swz<G>(p, ybase[pt] + ky)) * 16);""")
rep("""        f16x8 bA0[NPT], bA1[NPT], bA2[NPT], bB0[PF_ALL ? NPT : 1], bB1[PF_ALL ? NPT : 1], bB2[NPT];
        auto tap_body = [&](int tap, f16x8 (&b0)[NPT], f16x8 (&b1)[NPT], f16x8 (&b2)[NPT], auto& n0, auto& n1, f16x8 (&n2)[NPT]) {""","""        f16x8 bA0[NT], bA1[NT], bA2[NT], bB0[PF_ALL ? NT : 1], bB1[PF_ALL ? NT : 1], bB2[NT];
        auto tap_body = [&](int tap, f16x8 (&b0)[NT], f16x8 (&b1)[NT], f16x8 (&b2)[NT], auto& n0, auto& n1, f16x8 (&n2)[NT]) {""")
old_m="""#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a0, b2[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a2, b0[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a1, b1[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a1, b0[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a0, b1[pt], acc[ht][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ht][pt] = MFMA_H(a0, b0[pt], acc[ht][pt]);
            }
            ring_wait_barrier<0>();                                      // tap+1's weights landed; everyone is done with this tile"""
rep(old_m, old_m.replace("pt < NPT","pt < NT"))
rep("""                for (int pt = 0; pt < NPT; ++pt) sink += acc[ht][pt][0] + acc[ht][pt][1] + acc[ht][pt][2] + acc[ht][pt][3];
            if (sink == 1.2345e-30f) a.Q2[sn * F] = sink;""","""                for (int pt = 0; pt < NT; ++pt) sink += acc[ht][pt][0] + acc[ht][pt][1] + acc[ht][pt][2] + acc[ht][pt][3];
            if (sink == 1.2345e-30f) a.Q2[sn * F] = sink;""")
rep("""            if (hcb >= a.Hc) break;
            const f32x4 bias = *(const f32x4*)(a.K2b + (long long)sw * a.Hc + hcb + 4 * lg);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (pt * 16 + li < NPOS_) {
                        const float pre = acc[ht][pt][r] * out_scale + bias[r];   // sigmoid / tanh are pooled on their VALUES, as torch does
                        my[(4 * lg + r) * CPITCH + pt * 16 + li] = smooth_act<ACT>() ? act_fwd<ACT>(pre) : pre;
                    }
""","""            const bool valid = hcb < a.Hc;
            if (!PAIR && !valid) break;
            if (valid) {
            const f32x4 bias = *(const f32x4*)(a.K2b + (long long)sw * a.Hc + hcb + 4 * lg);
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) {
                const int c = c0 + pt * 16 + li, point = (PAIR && c >= NPOS_) ? 1 : 0, pos = c - point * NPOS_;
                float* const T = PAIR ? (float*)ldsb + (wq + 4 * point) * 16 * CPITCH : my;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (PAIR ? c < 2 * NPOS_ : pos < NPOS_) {
                        const float pre = acc[ht][pt][r] * out_scale + bias[r];   // sigmoid / tanh are pooled on their VALUES, as torch does
                        T[(4 * lg + r) * CPITCH + pos] = smooth_act<ACT>() ? act_fwd<ACT>(pre) : pre;
                    }
            }
            }
            if (PAIR) __syncthreads();
""")
rep("""                if (i4 < 16 * NP2_ / 4 && live) {""","""                if (i4 < 16 * NP2_ / 4 && live && valid) {""")
rep("""#endif
                }
            }
        }
        __syncthreads();                                                 // the pooling tiles alias the weight buffers of the next chunk
    }
}
""","""#endif
                }
            }
            if (PAIR && ht + 1 < HTW) __syncthreads();
        }
        __syncthreads();                                                 // the pooling tiles alias the weight buffers of the next chunk
    }
    };
    if (PAIR && wp) body(std::integral_constant<int, PAIR ? NPT - 1 : NPT>{});
    else body(std::integral_constant<int, NPT>{});
}
""")
open(sys.argv[1] if len(sys.argv)>1 else p,'w').write(s)
