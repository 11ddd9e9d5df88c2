// wgrad_zs.h -- z-streaming filter gradient of the 5^3 convolution on bf16 tensors (round 4; included by conv_b16.hip after
// conv_kernels.h).  Reference: the gradient op of layers2.py:59-63's tf.nn.conv3d w.r.t. its filter (model.py:660).
//
//   D[cout][cin] of tap t  +=  dy[voxel][cout]^T  x  x[voxel + t][cin]          (v_mfma_f32_16x16x32_bf16, k = 32 voxels)
//
// What the row-reuse kernel (conv_kernels.h: wgrad5_bf16_rr) leaves on the table at the deep levels, measured in round 4
// (profiles/r04_pmc_bf16.txt: matrix pipe 58 % busy): (1) it reads one x fragment + one dy fragment from LDS per FIVE MFMAs -- with
// two waves per SIMD the LDS is ~90 % busy; (2) every brick re-stages its whole halo (x tile 3.4x the brick) and (3) a workgroup owns
// a 16 x 16 channel block, so the same tiles are staged by every cout block.  Here:
//   * a workgroup owns 16 cin x 32 cout (two MFMA column blocks) and ALL 125 taps: wave w holds the three (dz, dx) pairs 3w..3w+2
//     with all five dy (+ one dy of the 25th pair on waves 0-4) = 32 accumulator tiles = 128 registers;
//   * per output row ALL of a wave's taps are served from one read of the row's two dy fragments and ONE new x row fragment per
//     pair (the pair's other four rows slide in registers): 6 fragment reads for 32 MFMAs, LDS ~40 % busy;
//   * a workgroup walks a COLUMN of the volume in z: one step = the 8 rows x TX voxels x ZP planes of one k-step plane group
//     (TX x ZP = 32: 32 x 1 at 32^3, 16 x 2 at 16^3, 8 x 4 at 8^3).  The x planes live in a ring of eight LDS slots (slot = input
//     plane & 7): a step stages only its ZP NEW planes + its dy tile (30 KB instead of the 140 KB of a haloed brick) -- 16-20
//     prefetch registers, which is what lets the 128 accumulator registers fit next to three sliding windows;
//   * lane group g of a k-step takes 4 + 4 voxels: x {4g..4g+3} and {16+4g..} of the row (TX = 32), x {4g..} of planes p and p + 1
//     (TX = 16), x {4(g&1)..} of planes (g>>1) and (g>>1) + 2 (TX = 8): every 32-lane half of a transpose read covers one
//     contiguous 256-byte run.
// Work split over workgroups: blockIdx -> (split, cin chunk, 32-cout block); the column steps of the layer are cut into nsplit
// contiguous ranges (z fastest), every range sums its steps in ascending order into one fp32 slab [tap][CinP][CoutP] (or dw
// itself when the layer is not split): deterministic, a function of (shape, nsplit) only.
#pragma once

namespace {

template <int TX>
struct ZsGeom {
    static constexpr int ZP = 32 / TX, TY = 8, IY = 12, IX = TX + 4, RING = TX == 8 ? 16 : 8;      // (TX = 8: D <= 12, the whole column stays)
    static constexpr int XROW = IX * 32, XPLANE = IY * XROW, XBYTES = RING * XPLANE;
    static constexpr int DPLANE = TY * TX * 32, DBLK = ZP * DPLANE, DYBYTES = 2 * DBLK;
    static constexpr size_t LDS = (size_t)XBYTES + 2 * DYBYTES;          // dy tile double-buffered
};

template <int TX>
__device__ __forceinline__ void wgrad5_b16_zs_body(const WgradArgs& a, const int bid_x, const int bid_y) {
    using Z = ZsGeom<TX>;
    constexpr int ZP = Z::ZP, TY = Z::TY, IY = Z::IY, IX = Z::IX, RM = Z::RING - 1;
    constexpr int XROW = Z::XROW, XPLANE = Z::XPLANE, DPLANE = Z::DPLANE, DBLK = Z::DBLK;
    constexpr int PU = IY * IX * 2;                                      // 16-byte units of one x plane
    constexpr int NXU = ZP * PU, KX = (NXU + 511) / 512;                 // the new planes of a step
    constexpr int NHU = 4 * PU, KH = (NHU + 511) / 512;                  // the four older planes at a column start
    constexpr int PSTEP = TX == 32 ? 0 : (TX == 16 ? 1 : 2);             // plane distance of a lane's second transpose read
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xt = smem;
    unsigned char* dyt = smem + Z::XBYTES;                               // two dy tiles: step parity
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int split = bid_x;
    const int chunk = bid_y / a.ncob, cob = bid_y - chunk * a.ncob;
    const int co0 = cob * 32;
    const int nzs = a.nbz;                                               // z steps per column

    // lane part of every transpose read
    const int lx = (TX == 8 ? 4 * (g & 1) : 4 * g) + (i >> 2);
    const int lp = TX == 8 ? (g >> 1) : 0;
    const int lq = (i & 3) * 8;
    // a fragment address = lane part (VGPR) + ring slot / tap part (wave-uniform) + row (immediate)
    const unsigned char* vx = xt + lp * XPLANE + lx * 32 + lq;            // (TX = 8: ring of 16, a column never wraps: the lane's plane is a constant offset)
    int dxc[3], dzc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int q = 3 * wave + c, dz = q / 5, dx = q - dz * 5;
        dxc[c] = dx * 32;
        dzc[c] = dz;
    }
    const bool extra = wave < 5;
    const int xe = 4 * 32 + (extra ? wave : 0) * XROW;                     // pair (dz, dx) = (4, 4): tap dy = wave on waves 0-4
    const unsigned char* pa = dyt + lp * DPLANE + lx * 32 + lq;

    f32x4 acc[16][2];
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int c16 = chunk * 16;
    const bool first_src = c16 < a.C0;
    const int cs = first_src ? a.C0 : a.C1;
    const unsigned short* xsrc = first_src ? reinterpret_cast<const unsigned short*>(a.x0) + c16
                                           : reinterpret_cast<const unsigned short*>(a.x1) + (c16 - a.C0);
    const unsigned short* dsrc = reinterpret_cast<const unsigned short*>(a.dy);
    const size_t xvol = (size_t)a.Di * a.Hi * a.Wi * cs, dvol = (size_t)a.Do * a.Ho * a.Wo * a.Cout;
    const int Cin = a.C0 + a.C1;

    auto item_coords = [&](int item, int& b, int& by, int& bx, int& zb) {
        const int col = item / nzs;
        zb = (item - col * nzs) * ZP;
        bx = col % a.nbx;
        const int t = col / a.nbx;
        by = t % a.nby; b = t / a.nby;
    };
    // one 16-byte unit (plane p of a plane group that starts at input plane z0, row, voxel, channel half) of the x tile
    auto x_unit = [&](int e, int& p, int& row, int& ix, int& hf) {
        p = e / PU;
        const int r = e - p * PU;
        row = r / (IX * 2);
        const int c2 = r - row * (IX * 2);
        ix = c2 >> 1; hf = c2 & 1;
    };
    auto x_load = [&](const unsigned short* src, int e, int nunits, int z0, int by, int bx) -> u32x4 {
        int p, row, ix, hf;
        x_unit(e, p, row, ix, hf);
        const int gz = z0 + p, gy = by * TY - 2 + row, gx = bx * TX - 2 + ix;
        const bool ok = e < nunits && (unsigned)gz < (unsigned)a.Di && (unsigned)gy < (unsigned)a.Hi && (unsigned)gx < (unsigned)a.Wi &&
                        c16 + hf * 8 < Cin;
        const unsigned off = ok ? (unsigned)(((gz * a.Hi + gy) * a.Wi + gx) * cs + hf * 8) : 0u;
        return load16_or_zero(src + off, ok);
    };
    auto x_store = [&](int e, int nunits, int u0, const u32x4& v) {            // u0: ring coordinate (input plane + 2) of plane 0
        int p, row, ix, hf;
        x_unit(e, p, row, ix, hf);
        if (e < nunits) *reinterpret_cast<u32x4*>(xt + ((u0 + p) & RM) * XPLANE + row * XROW + ix * 32 + hf * 16) = v;
    };

    u32x4 hx[KX], hd[2];
    unsigned xdst[KX];                             // LDS byte offsets of hx[] (computed with the loads, while registers are free); ~0u = none
    auto issue = [&](int item) {
        int b, by, bx, zb;
        item_coords(item, b, by, bx, zb);
        int tv = tid;
        asm volatile("" : "+v"(tv));               // (keeps the per-thread unit arithmetic inside the step: hoisted, it holds registers)
        const unsigned short* src = xsrc + (size_t)b * xvol;
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            hx[k] = x_load(src, tv + k * 512, NXU, zb + 2, by, bx);                                // input planes zb + 2 .. zb + ZP + 1
            int p, row, ix, hf;
            x_unit(tv + k * 512, p, row, ix, hf);
            xdst[k] = tv + k * 512 < NXU ? (unsigned)(((zb + 4 + p) & RM) * XPLANE + row * XROW + ix * 32 + hf * 16) : ~0u;
        }
        const unsigned short* dsr = dsrc + (size_t)b * dvol;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = tv + k * 512;
            const int v = e >> 2, cu = e & 3;
            const int p = v / (TY * TX), y = (v / TX) % TY, x = v % TX;
            const int oz = zb + p, oy = by * TY + y, ox = bx * TX + x, cd = co0 + (cu >> 1) * 16 + (cu & 1) * 8;
            const bool ok = oz < a.Do && oy < a.Ho && ox < a.Wo && cd < a.Cout;
            const unsigned off = ok ? (unsigned)(((oz * a.Ho + oy) * a.Wo + ox) * a.Cout + cd) : 0u;
            hd[k] = load16_or_zero(dsr + off, ok);
        }
    };
    auto commit = [&](int par) {
        int tv = tid;
        asm volatile("" : "+v"(tv));
#pragma unroll
        for (int k = 0; k < KX; ++k) if (xdst[k] != ~0u) *reinterpret_cast<u32x4*>(xt + xdst[k]) = hx[k];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = tv + k * 512;
            const int v = e >> 2, cu = e & 3;
            *reinterpret_cast<u32x4*>(dyt + par * Z::DYBYTES + (cu >> 1) * DBLK + v * 32 + (cu & 1) * 16) = hd[k];
        }
    };
    auto halo = [&](int item) {                    // column start: the four input planes zb - 2 .. zb + 1 (ring coordinates zb .. zb + 3)
        int b, by, bx, zb;
        item_coords(item, b, by, bx, zb);
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const unsigned short* src = xsrc + (size_t)b * xvol;
        u32x4 hh[KH];
#pragma unroll
        for (int k = 0; k < KH; ++k) hh[k] = x_load(src, tv + k * 512, NHU, zb - 2, by, bx);
#pragma unroll
        for (int k = 0; k < KH; ++k) x_store(tv + k * 512, NHU, zb, hh[k]);
    };
    auto tr2 = [&](const unsigned char* p0, const unsigned char* p1) -> bf16x8 {
#ifdef ZS_NO_READS
        bf16x8 z;
        asm volatile("" : "=v"(z) : "v"(p0), "v"(p1));
        return z;
#endif
        typedef s16x4 __attribute__((address_space(3))) * lp3;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp3)p0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp3)p1);
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    const int nitems = a.nbrick;
    const int i0 = (int)((long long)split * nitems / a.nsplit), i1 = (int)((long long)(split + 1) * nitems / a.nsplit);
    // Pipeline: ONE barrier per step.  The ring has spare slots (8 - (ZP + 4) >= ZP; 16 - 8 at TX = 8) and the dy tile two buffers, so
    // the tiles of step s + 1 are loaded into registers at the head of step s and written to LDS in the MIDDLE of step s (after
    // row 3: four rows of MFMAs cover the loads, nobody reads those slots during step s); the barrier at the head of step s + 1
    // publishes them.  Only a column change (the new column's planes would overwrite live slots) takes the old path: barrier,
    // load + write, barrier.
#ifndef ZS_NO_STAGE
    if (i0 < i1) {
        int b, by, bx, zb;
        item_coords(i0, b, by, bx, zb);
        issue(i0);
        halo(i0);
        commit(0);
    }
#endif
    for (int item = i0; item < i1; ++item) {
        int b, by, bx, zb;
        item_coords(item, b, by, bx, zb);
        const int par = (item - i0) & 1;
        const bool next = item + 1 < i1;
        const bool samecol = next && zb + ZP < nzs * ZP;                     // (the next step continues this column)
        __syncthreads();                           // this step's tiles are in LDS; every wave is done with the previous step
#ifndef ZS_NO_STAGE
        if (next) {
            issue(item + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        // this step's wave-uniform fragment offsets: ring slot of (output plane group zb [+ PSTEP] + dz) + the tap's dx
        int sx0[3], sx1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sx0[c] = ((zb + dzc[c]) & RM) * XPLANE + dxc[c];
            sx1[c] = TX == 32 ? sx0[c] + 512 : ((zb + PSTEP + dzc[c]) & RM) * XPLANE + dxc[c];
        }
        const int se0 = ((zb + 4) & RM) * XPLANE + xe;
        const int se1 = TX == 32 ? se0 + 512 : ((zb + PSTEP + 4) & RM) * XPLANE + xe;
        constexpr int A1OFF = TX == 32 ? 512 : PSTEP * DPLANE;
        const unsigned char* pap = pa + par * Z::DYBYTES;
        // (the offsets are laundered per row where registers are short, so that hipcc forms vx + offset next to each read instead
        //  of holding eight more pointers across the step: TX = 16 / 8 spilled 12 / 37 VGPRs with them)
        auto fx = [&](int o0, int o1, int row) -> bf16x8 {
            if constexpr (TX != 32) { asm volatile("" : "+s"(o0)); asm volatile("" : "+s"(o1)); }
            return tr2(vx + o0 + row * XROW, vx + o1 + row * XROW);
        };

        // Two passes over the step's eight rows: pairs 0 and 1 (20 MFMAs per row), then pair 2 and the 25th pair's tap (12 MFMAs per
        // row).  All three windows at once need 60 registers next to the 128 accumulators, the prefetched tiles and the dy fragments:
        // hipcc then keeps the PREFETCHED TILES in scratch (measured: 52-120 bytes per lane in every arrangement tried).  Between the
        // passes no window is live: that is where the next step's tiles are written to LDS.  Price: the dy fragments are read twice.
        auto rows = [&](auto c0_, auto c1_, auto ex_) {
            constexpr int C0 = decltype(c0_)::value, C1 = decltype(c1_)::value;
            constexpr bool EX = decltype(ex_)::value;
            bf16x8 F[C1 - C0][5];
#pragma unroll
            for (int c = C0; c < C1; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) F[c - C0][r] = fx(sx0[c], sx1[c], r);
            bf16x8 An[2] = {tr2(pap, pap + A1OFF), tr2(pap + DBLK, pap + DBLK + A1OFF)};
#pragma unroll
            for (int y = 0; y < TY; ++y) {
                const bf16x8 A0 = An[0], A1 = An[1];
                // the row that slides in (needed by the dy = 4 taps only, which come last), the 25th pair's fragment, the next row's dy
#pragma unroll
                for (int c = C0; c < C1; ++c) F[c - C0][(y + 4) % 5] = fx(sx0[c], sx1[c], y + 4);
                bf16x8 E = A0;
                if (EX && extra) E = fx(se0, se1, y);                                        // (wave-uniform)
                if (y + 1 < TY) { An[0] = tr2(pap + (y + 1) * (TX * 32), pap + (y + 1) * (TX * 32) + A1OFF);
                                  An[1] = tr2(pap + DBLK + (y + 1) * (TX * 32), pap + DBLK + (y + 1) * (TX * 32) + A1OFF); }
                __builtin_amdgcn_sched_barrier(0);
#ifdef ZS_NO_MFMA
#pragma unroll
                for (int c = C0; c < C1; ++c) asm volatile("" :: "v"(F[c - C0][(y + 4) % 5]), "v"(A0), "v"(A1), "v"(E));
                if (zb < 0)
#endif
                {
#pragma unroll
                for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                    for (int c = C0; c < C1; ++c) {
                        acc[c * 5 + dy][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, F[c - C0][(y + dy) % 5], acc[c * 5 + dy][0], 0, 0, 0);
                        acc[c * 5 + dy][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, F[c - C0][(y + dy) % 5], acc[c * 5 + dy][1], 0, 0, 0);
                    }
                if (EX && extra) {
                    acc[15][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, E, acc[15][0], 0, 0, 0);
                    acc[15][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, E, acc[15][1], 0, 0, 0);
                }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        rows(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::false_type{});
#ifndef ZS_NO_STAGE
        if (samecol) {                             // (uniform) the next step's tiles: spare ring slots, the other dy buffer
            commit(par ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        rows(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, std::true_type{});
#ifndef ZS_NO_STAGE
        if (next && !samecol) {                    // column change: the new column's planes go where this step's planes still were
            __syncthreads();
            halo(item + 1);
            commit(par ^ 1);
        }
#endif
    }
    // lane holds dW[tap][ci = chunk*16 + i][co = co0 + n*16 + 4*g + {0..3}]
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        int tap;
        if (t < 15) { const int q = 3 * wave + t / 5, dz = q / 5, dx = q - dz * 5; tap = (dz * 5 + t % 5) * 5 + dx; }
        else { if (!extra) continue; tap = (4 * 5 + wave) * 5 + 4; }
        float* dst = a.part + ((size_t)(split * 125 + tap) * a.CinP + c16 + i) * a.CoutP + co0 + g * 4;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const f32x4 r = acc[t][n];
            *reinterpret_cast<float4*>(dst + n * 16) = make_float4(r.x, r.y, r.z, r.w);
        }
    }
}

template <int TX>
__global__ void __launch_bounds__(512) wgrad5_b16_zs_kernel(WgradArgs a) {
    wgrad5_b16_zs_body<TX>(a, blockIdx.x, blockIdx.y);
}

// shapes the z-streaming kernel takes: whole 32-cout blocks, 16-cin chunks that do not straddle the two sources
inline bool zs_shape_ok(int C0, int C1, int Cout) {
    return (round_up(Cout, 16) % 32) == 0 && !(C0 & 7) && !(C1 & 7) && !(Cout & 7) && (C1 == 0 || (C0 & 15) == 0);
}
inline int zs_tx(int W) { return W >= 32 ? 32 : (W >= 16 ? 16 : 8); }
// (the 8-wide form keeps the whole column's planes in LDS: 16 ring slots)
inline bool zs_depth_ok(int D, int W) { return W >= 16 || D <= 12; }

// fills the geometry fields of `a` (columns, z steps); returns the number of column steps of one (chunk, cout block)
inline int zs_geometry(WgradArgs& a) {
    const int tx = zs_tx(a.Wo), zp = 32 / tx;
    a.ncob = a.CoutP / 32;
    a.nbz = ceil_div(a.Do, zp); a.nby = ceil_div(a.Ho, 8); a.nbx = ceil_div(a.Wo, tx);
    a.nbrick = a.B * a.nby * a.nbx * a.nbz;
    return a.nbrick;
}

template <int TX>
int launch_wgrad_zs_t(const WgradArgs& a, hipStream_t st) {
    auto k = wgrad5_b16_zs_kernel<TX>;
    static unsigned long long attr_done = 0;
    if (int ae = ensure_lds(k, ZsGeom<TX>::LDS, attr_done)) return ae;
    dim3 grid(a.nsplit, (a.CinP / 16) * a.ncob, 1);
    hipLaunchKernelGGL(k, grid, dim3(512), ZsGeom<TX>::LDS, st, a);
    return (int)hipGetLastError();
}
inline int launch_wgrad_zs(const WgradArgs& a, hipStream_t st) {
    const int tx = zs_tx(a.Wo);
    return tx == 32 ? launch_wgrad_zs_t<32>(a, st) : tx == 16 ? launch_wgrad_zs_t<16>(a, st) : launch_wgrad_zs_t<8>(a, st);
}

}  // namespace
