// fused_sweep_alt_kernels.hpp — the measured-and-rejected kernel forms of the fused sweep (profiles/NOTES.md):
// the whole-cycle kernels k_cycle_xy / k_cycle_pc and the LDS-transposed X march k_sweep_x_lds. Correct and tested, 1.3-4x slower
// than what the solver runs. Compiled only with -DARMON_ALT_KERNELS, into libarmon_hip_alt.so (build.py). This file is
// included by fused_sweep_impl.hpp INSIDE its anonymous namespace, after the product kernels it shares helpers with
// (sweep_args, buf_load / buf_store, cfl_track, bc_source, static_for): it is not a stand-alone header.

// ---- whole cycle X then Y in ONE pass over memory (Sequential splitting) -----------------------------------------
// A wave owns 64 consecutive columns (56 produced + the X sweep's 4-cell halo on both sides) and marches down y:
// each step loads one row segment, sweeps it along x in place (lanes along x, DPP shifts: SpatialSweep<K = 1>) and
// feeds the X-swept cell of every lane straight into that lane's Y pipeline (the register march of k_sweep_y): the
// intermediate state between the two sweeps of the reference's cycle (ref src/solver.jl:300-316 executed for X,
// then for Y) never leaves the registers. 32 B read + 32 B written per cell per CYCLE instead of per sweep.
#ifndef ARMON_C_PF
#define ARMON_C_PF 3
#endif
template <int SCHEME, int LIM, int PROJ, int EOS, bool EXACT, bool TRACK>
#ifndef ARMON_C_WAVES
#define ARMON_C_WAVES 2
#endif
__global__ void __launch_bounds__(256, ARMON_C_WAVES)
k_cycle_xy(sweep_args a, real dt_y, real dx_y)
{
    using SW = fused::SpatialSweep<SCHEME, LIM, PROJ, EOS, EXACT, 1, real>;
    using PIPE = typename std::conditional<EXACT, fused::Pipe<SCHEME, LIM, PROJ, EOS, real>, fused::PipeFast<SCHEME, LIM, PROJ, EOS, real>>::type;
    using St = fused::Strip<1, real>;
    constexpr int LAG = PIPE::LAG, PF = ARMON_C_PF, VALID = 64 - 2 * LAG;
    const int nx = (int)a.nx, ny = (int)a.ny, g = a.g;
    const int lane = threadIdx.x & 63, wave = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int xr = wave * VALID - LAG + lane;                 // this lane's column (may be a ghost: -LAG .. nx + LAG - 1)
    const bool produces = lane >= LAG && lane < 64 - LAG && xr < nx;
    // X boundary: mirror of the inside (ref src/halo_exchange.jl:2-29) or ghost columns filled by the halo exchange
    int xs = xr < -g ? -g : (xr > nx + g - 1 ? nx + g - 1 : xr);
    real fxa = 1, fxt = 1;
    if (xs < 0 && a.bc_low) { fxa = a.fa_low; fxt = a.ft_low; xs = -1 - xs; }
    else if (xs >= nx && a.bc_high) { fxa = a.fa_high; fxt = a.ft_high; xs = 2 * nx - 1 - xs; }
    const int o_hi = (int)a.o_hi;
    const int o0 = (int)a.o_lo + (int)blockIdx.y * a.seg;
    const int o1 = (o0 + a.seg < o_hi) ? o0 + a.seg : o_hi;
    const int jb = o0 - LAG, je = o1 + LAG;

    const unsigned colb = (unsigned)(xs + g) * (unsigned)sizeof(real), colw = (unsigned)(xr + g) * (unsigned)sizeof(real);
    const unsigned pitchb = (unsigned)a.row_len * (unsigned)sizeof(real);
    const int64_t in_base = (int64_t)(jb + g) * a.row_len, out_base = (int64_t)(o0 + g) * a.row_len;
    // X sweep: ua = u, ut = v
    const rsrc_t r_rho = make_rsrc(a.rho_in + in_base), r_u = make_rsrc(a.ua_in + in_base);
    const rsrc_t r_v = make_rsrc(a.ut_in + in_base), r_E = make_rsrc(a.E_in + in_base);
    const rsrc_t w_rho = make_rsrc(a.rho_out + out_base), w_u = make_rsrc(a.ua_out + out_base);
    const rsrc_t w_v = make_rsrc(a.ut_out + out_base), w_E = make_rsrc(a.E_out + out_base);

    SW sw{a.dt, a.dx, a.gamma, a.inv_dx, a.dt_dx};
    PIPE pipe(dt_y, dx_y, a.gamma);
    cfl_track cfl;
    // Y boundary factors act on the X-swept state exactly as the reference's mirror does between its two sweeps
    St raw[PF + 1][4];
    int lj = jb;
    unsigned so_off = (unsigned)(jb - LAG - o0) * pitchb;
    auto load = [&](auto slot) {
        constexpr int K = decltype(slot)::value % (PF + 1);
        const bool m_lo = lj < 0 && a.bc_low_t, m_hi = lj >= ny && a.bc_high_t;
        const int src = m_lo ? -1 - lj : (m_hi ? 2 * ny - 1 - lj : lj);
        const unsigned off = (unsigned)(src - jb) * pitchb;
        // transverse (y) mirror: v flips with the Y factor, u with its own
        const real fu = (m_lo ? a.tu_low : (m_hi ? a.tu_high : real(1))) * fxa;
        const real fv = (m_lo ? a.tv_low : (m_hi ? a.tv_high : real(1))) * fxt;
        raw[K][0].v[0] = buf_load<real>(r_rho, colb, off);
        raw[K][1].v[0] = buf_load<real>(r_u, colb, off) * fu;
        raw[K][2].v[0] = buf_load<real>(r_v, colb, off) * fv;
        raw[K][3].v[0] = buf_load<real>(r_E, colb, off);
        if (lj + 1 < je) lj++;
    };
    auto step = [&](auto ph, int j) {
        constexpr int PH = decltype(ph)::value;              // multiple-of-(8·(PF+1)) unroll: both rings by phase
        constexpr int PH8 = PH & 7, KR = PH % (PF + 1);
        St o_rho, o_u, o_v, o_E, p, cs;
        sw.run(raw[KR][0], raw[KR][1], raw[KR][2], raw[KR][3], o_rho, o_u, o_v, o_E, p, cs);
        load(std::integral_constant<int, PH + PF + 1>{});     // row j + PF + 1 → the slot just consumed
        real pj, cj, c_lag;
        const fused::Out4<real> out = pipe.template push<true, PH8>(o_rho.v[0], o_v.v[0], o_u.v[0], o_E.v[0], pj, cj, c_lag);
        const int o = j - LAG;
        if (a.emit && j >= o0 && j < o1 && produces) {
            const unsigned off = so_off + LAG * pitchb;
            if (a.emit & 1) buf_store(make_rsrc(a.p_out + out_base), colw, off, pj);
        }
        if (o >= o0 && o < o1) {
            if (produces) {
                buf_store(w_rho, colw, so_off, out.rho);
                buf_store(w_u, colw, so_off, out.ut);         // Y pipeline: ut = u, ua = v
                buf_store(w_v, colw, so_off, out.ua);
                buf_store(w_E, colw, so_off, out.E);
                if (TRACK) cfl.add(out.ut, out.ua, c_lag);
            }
        }
        so_off += pitchb;
    };
    constexpr int U = 8 * (PF + 1);
    static_for(std::make_integer_sequence<int, PF + 1>{}, [&](auto k) { load(k); });
    const int T = je - jb;
    for (int t = 0; t < T; t += U)
        static_for(std::make_integer_sequence<int, U>{}, [&](auto ph) {
            if (t + decltype(ph)::value < T) step(ph, jb + t + decltype(ph)::value);
        });
    if (TRACK) cfl_block_store<4>(cfl, a.partials, (int64_t)blockIdx.y * gridDim.x + blockIdx.x, threadIdx.x);
}

// ---- whole cycle, producer / consumer waves (x_kernel = 4; measured alternative) -------------------------------------
// k_cycle_xy above needs the registers of BOTH stages in every wave (≈340: one wave per SIMD, VALU-bound). Here the
// stages live in different waves of a 3-wave workgroup and meet in LDS:
//   wave 0 (producer)  : X stage of one row of a 128-cell strip (2 cells per lane, DPP shifts, the X sweep's own code),
//                        120 X-swept cells written to a two-row LDS ring (4 KB per row);
//   waves 1, 2 (consumers): each marches 64 / 56 of those columns down y with the Y sweep's register pipeline, reading
//                        the row the producer finished in the previous iteration.
// One workgroup barrier per row hands a row over (iteration t: the producer works on row t while the consumers work on
// row t-1; the slot written in iteration t+1 is the one read in iteration t-1). Bit-identical to the two sweeps, and
// the VALU work per cell is theirs while the HBM traffic halves — but a kernel has ONE register allocation: hipcc
// (ROCm 7.2) does not overlay the two roles' registers (143 alone + 166 alone -> 256 + 136 B of scratch inside the row
// loops, whether the roles are branches of one loop, separate loops, scoped blocks or — unconstrained — non-inlined
// functions at 248 / 256), and with the spills it runs at 11.5 ms per cycle against 5.7 ms for the two launches.
// Kept selectable for that record; it needs either a register budget per role (hand-written assembly) or a march
// whose rings live in LDS.
constexpr int kPcValid = 120;

// geometry of one workgroup of k_cycle_pc, shared by its two roles
struct pc_geom { int64_t w0; int o0, o1, jb, T; };
constexpr int kPcWidth = 128, kPcHalo = 4;
typedef real pc_ring_t[2][4][kPcWidth];

// The roles are separate NON-INLINED functions: inlined into one kernel body, the register allocator adds the two
// roles' registers up (143 + 166 -> 256 VGPRs + scratch); as calls, the kernel needs the larger of the two.
template <int SCHEME, int LIM, int PROJ, int EOS>
__device__ __forceinline__ void pc_producer(const sweep_args& a, const pc_geom& gm, pc_ring_t& ring)
{
    using SW = fused::SpatialSweep<SCHEME, LIM, PROJ, EOS, false, 2, real>;
    using St = fused::Strip<2, real>;
    constexpr int HALO = kPcHalo, WIDTH = kPcWidth;
    const int lane = threadIdx.x & 63;
    const int64_t nx = a.nx, w0 = gm.w0;
    const int ny = (int)a.ny, g = a.g, jb = gm.jb, T = gm.T;
        // ---- producer: X stage of row jb + t into ring slot t & 1
        SW sw{a.dt, a.dx, a.gamma, a.inv_dx, a.dt_dx};
        St buf[2][4];
        const int64_t cb = w0 - HALO;                                    // first cell of the strip
        const int64_t j0 = cb + 2 * (int64_t)lane;                       // this lane's first cell
        const bool interior = cb >= 0 && cb + WIDTH <= nx;               // no ghost column, no clamping
        const bool vec_ok = (a.row_len % 2 == 0) && ((cb + g) % 2 == 0);
        auto load_row = [&](auto slot, int t) {
            constexpr int B = decltype(slot)::value;
            const int j = jb + t;
            const bool m_lo = j < 0 && a.bc_low_t, m_hi = j >= ny && a.bc_high_t;    // y boundary: mirror rows
            const int src_row = m_lo ? -1 - j : (m_hi ? 2 * ny - 1 - j : j);
            const real fu = m_lo ? a.tu_low : (m_hi ? a.tu_high : real(1));
            const real fv = m_lo ? a.tv_low : (m_hi ? a.tv_high : real(1));
            const int64_t row_off = ((int64_t)src_row + g) * a.row_len + g;
            const real* in[4] = {a.rho_in + row_off, a.ua_in + row_off, a.ut_in + row_off, a.E_in + row_off};
            St& rho = buf[B][0];
            St& u = buf[B][1];
            St& v = buf[B][2];
            St& E = buf[B][3];
            if (interior && vec_ok) {
                const vec2 r = ld2(in[0] + j0), uu = ld2(in[1] + j0), vv = ld2(in[2] + j0), e = ld2(in[3] + j0);
                rho.v[0] = r.x; rho.v[1] = r.y;
                u.v[0] = uu.x * fu; u.v[1] = uu.y * fu;
                v.v[0] = vv.x * fv; v.v[1] = vv.y * fv;
                E.v[0] = e.x; E.v[1] = e.y;
            } else {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    int64_t jc = j0 + k;
                    jc = jc < -(int64_t)g ? -(int64_t)g : (jc > nx + g - 1 ? nx + g - 1 : jc);
                    real fa, ft;
                    const int64_t src = bc_source(a, nx, jc, fa, ft);    // x boundary: mirror columns
                    rho.v[k] = in[0][src];
                    u.v[k] = in[1][src] * fa * fu;
                    v.v[k] = in[2][src] * ft * fv;
                    E.v[k] = in[3][src];
                }
            }
        };
        load_row(std::integral_constant<int, 0>{}, 0);
        for (int t0 = 0; t0 <= T; t0 += 2)
            static_for(std::make_integer_sequence<int, 2>{}, [&](auto ph) {
                constexpr int B = decltype(ph)::value;
                const int t = t0 + B;
                if (t <= T) {
                    if (t < T) {
                        if (t + 1 < T) load_row(std::integral_constant<int, 1 - B>{}, t + 1);   // next row in flight during this one
                        St o_rho, o_u, o_v, o_E, p, cs;
                        sw.run(buf[B][0], buf[B][1], buf[B][2], buf[B][3], o_rho, o_u, o_v, o_E, p, cs);
                        vreal2* dst = reinterpret_cast<vreal2*>(&ring[B][0][0]);
                        dst[0 * (WIDTH / 2) + lane] = vreal2{o_rho.v[0], o_rho.v[1]};
                        dst[1 * (WIDTH / 2) + lane] = vreal2{o_u.v[0], o_u.v[1]};
                        dst[2 * (WIDTH / 2) + lane] = vreal2{o_v.v[0], o_v.v[1]};
                        dst[3 * (WIDTH / 2) + lane] = vreal2{o_E.v[0], o_E.v[1]};
                    }
                    __syncthreads();
                }
            });
}

template <int SCHEME, int LIM, int PROJ, int EOS, bool TRACK>
__device__ __forceinline__ void pc_consumer(const sweep_args& a, const pc_geom& gm, pc_ring_t& ring, real dt_y, real dx_y,
                                                      int role, cfl_track& cfl)
{
    using PIPE = fused::PipeFast<SCHEME, LIM, PROJ, EOS, real>;
    constexpr int LAG = PIPE::LAG, HALO = kPcHalo;
    const int lane = threadIdx.x & 63;
    const int64_t nx = a.nx, w0 = gm.w0;
    const int g = a.g, jb = gm.jb, T = gm.T, o0 = gm.o0, o1 = gm.o1;
    // keep the march's initial state from being hoisted above the role branch (it would stay live through the producer)
    asm volatile("" : "+v"(dt_y), "+v"(dx_y));
        // ---- consumers: wave 1 marches produced cells 0..63, wave 2 cells 64..119; row t - 1 from ring slot (t - 1) & 1
        PIPE pipe(dt_y, dx_y, a.gamma);
        const int pc = (role == 2 ? 64 : 0) + lane;                      // produced cell of this lane
        const int ci = HALO + (pc < kPcValid ? pc : kPcValid - 1);       // its place in the strip
        const int64_t xr = w0 + pc;
        const bool active = pc < kPcValid && xr >= 0 && xr < nx;
        const unsigned colw = (unsigned)((active ? xr : 0) + g) * (unsigned)sizeof(real);
        const unsigned pitchb = (unsigned)a.row_len * (unsigned)sizeof(real);
        const int64_t out_base = (int64_t)(o0 + g) * a.row_len;
        const rsrc_t w_rho = make_rsrc(a.rho_out + out_base), w_u = make_rsrc(a.ua_out + out_base);
        const rsrc_t w_v = make_rsrc(a.ut_out + out_base), w_E = make_rsrc(a.E_out + out_base);
        for (int t0 = 0; t0 <= T; t0 += 8)
            static_for(std::make_integer_sequence<int, 8>{}, [&](auto ph) {
                constexpr int PH = decltype(ph)::value, PH8 = (PH + 7) & 7, B = PH8 & 1;     // phase and ring slot of row t - 1
                const int t = t0 + PH;
                if (t <= T) {
                    if (t >= 1) {
                        const real rho = ring[B][0][ci], u = ring[B][1][ci], v = ring[B][2][ci], E = ring[B][3][ci];
                        real pj, cj, c_lag;
                        const fused::Out4<real> out = pipe.template push<true, PH8>(rho, v, u, E, pj, cj, c_lag);   // Y march: ua = v, ut = u
                        const int j = jb + t - 1, o = j - LAG;
                        if ((a.emit & 1) && j >= o0 && j < o1 && active)
                            buf_store(make_rsrc(a.p_out + out_base), colw, (unsigned)(j - o0) * pitchb, pj);
                        if (o >= o0 && o < o1 && active) {
                            const unsigned so_off = (unsigned)(o - o0) * pitchb;
                            buf_store(w_rho, colw, so_off, out.rho);
                            buf_store(w_u, colw, so_off, out.ut);
                            buf_store(w_v, colw, so_off, out.ua);
                            buf_store(w_E, colw, so_off, out.E);
                            if (TRACK) cfl.add(out.ut, out.ua, c_lag);
                        }
                    }
                    __syncthreads();
                }
            });
}

template <int SCHEME, int LIM, int PROJ, int EOS, bool TRACK>
#ifndef ARMON_PC_WAVES
#define ARMON_PC_WAVES 2
#endif
__global__ void __launch_bounds__(192, ARMON_PC_WAVES)
k_cycle_pc(sweep_args a, real dt_y, real dx_y)
{
    constexpr int LAG = fused::PipeTraits<SCHEME, LIM, PROJ, EOS>::LAG;
    static_assert(LAG <= kPcHalo, "strip halo");
    __shared__ __attribute__((aligned(16))) pc_ring_t ring;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform, and the compiler is told so
    pc_geom gm;
    gm.w0 = a.x_first + (int64_t)blockIdx.x * kPcValid;      // first column this workgroup produces (may be < 0)
    gm.o0 = (int)a.o_lo + (int)blockIdx.y * a.seg;
    gm.o1 = (gm.o0 + a.seg < (int)a.o_hi) ? gm.o0 + a.seg : (int)a.o_hi;
    gm.jb = gm.o0 - LAG;
    gm.T = gm.o1 + LAG - gm.jb;                               // rows to X-sweep
    cfl_track cfl;
    // both roles run T + 1 barriers
    if (role == 0) pc_producer<SCHEME, LIM, PROJ, EOS>(a, gm, ring);
    else pc_consumer<SCHEME, LIM, PROJ, EOS, TRACK>(a, gm, ring, dt_y, dx_y, role, cfl);
    if (TRACK) cfl_block_store<3>(cfl, a.partials, (int64_t)blockIdx.y * gridDim.x + blockIdx.x, threadIdx.x);
}

// ---- X sweep, LDS-transposed march (alternative form) -----------------------------------------------------
constexpr int kXRows = 64;      // one wave: lane ↔ row
constexpr int kXChunk = 8;

template <class PIPE, int CH, bool TRACK>
__global__ void __launch_bounds__(kXRows, 2)
k_sweep_x_lds(sweep_args a)
{
    constexpr int LAG = PIPE::LAG;
    constexpr int PITCH = CH + 1;                 // odd pitch in doubles: conflict-free column walks
    constexpr int RPI = kXRows / CH;              // rows covered by one wave-wide row-segment access
    extern __shared__ real tile[];              // [planes][64][PITCH], planes = 4 (+2 when emitting p, c)

    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * kXRows;
    const int64_t o0 = a.o_lo + (int64_t)blockIdx.x * a.seg;
    const int64_t o1 = (o0 + a.seg < a.o_hi) ? o0 + a.seg : a.o_hi;
    const int64_t j_end = o1 + LAG;
    const bool row_ok = (r0 + lane) < a.ny;
    const bool emit = a.emit != 0;

    const int sub_row = lane / CH, sub_col = lane % CH;   // row-segment phase: lane → (row in group, column)
    auto T = [&](int plane, int r, int t) -> real& { return tile[(plane * kXRows + r) * PITCH + t]; };

    PIPE pipe(a.dt, a.dx, a.gamma);
    cfl_track cfl;

    for (int64_t jb = o0 - LAG; jb < j_end; jb += CH) {
        // -- load phase: 64 rows × CH columns, 8 B per lane, CH*8 B contiguous per row
        {
            const int64_t j = jb + sub_col;
            real fa, ft;
            const int64_t src = bc_source(a, a.nx, j, fa, ft);
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int r = k * RPI + sub_row;
                const int64_t row = r0 + r;
                if (row < a.ny && j < j_end) {
                    const int64_t idx = (row + a.g) * a.row_len + (src + a.g);
                    T(0, r, sub_col) = a.rho_in[idx];
                    T(1, r, sub_col) = a.ua_in[idx] * fa;
                    T(2, r, sub_col) = a.ut_in[idx] * ft;
                    T(3, r, sub_col) = a.E_in[idx];
                }
            }
        }
        __syncthreads();
        // -- march: lane walks its own row through the tile, results overwrite the consumed slots
        if (row_ok) {
            static_assert(CH == 8, "the march is unrolled by the 8 slots of the pipeline's cell ring");
            static_for(std::make_integer_sequence<int, 8>{}, [&](auto phc) {
                constexpr int t = decltype(phc)::value;
                const int64_t j = jb + t;
                if (j < j_end) {
                    real p, c, c_lag;
                    const fused::Out4<real> out = pipe.template push<false, t>(T(0, lane, t), T(1, lane, t), T(2, lane, t), T(3, lane, t), p, c, c_lag);
                    T(0, lane, t) = out.rho;
                    T(1, lane, t) = out.ua;
                    T(2, lane, t) = out.ut;
                    T(3, lane, t) = out.E;
                    if (emit) {
                        T(4, lane, t) = p;
                        T(5, lane, t) = c;
                    }
                    if (TRACK && j - LAG >= o0) cfl.add(out.ua, out.ut, c_lag);
                }
            });
        }
        __syncthreads();
        // -- store phase: slot t holds the new state of column jb + t - LAG (and p, c of column jb + t)
        {
            const int64_t j = jb + sub_col;
            const int64_t o = j - LAG;
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int r = k * RPI + sub_row;
                const int64_t row = r0 + r;
                if (row < a.ny && j < j_end) {
                    if (o >= o0) {
                        const int64_t io = (row + a.g) * a.row_len + (o + a.g);
                        a.rho_out[io] = T(0, r, sub_col);
                        a.ua_out[io] = T(1, r, sub_col);
                        a.ut_out[io] = T(2, r, sub_col);
                        a.E_out[io] = T(3, r, sub_col);
                    }
                    if (emit && j >= o0 && j < o1) {
                        const int64_t ij = (row + a.g) * a.row_len + (j + a.g);
                        if (a.emit & 1) a.p_out[ij] = T(4, r, sub_col);
                        if (a.emit & 2) a.c_out[ij] = T(5, r, sub_col);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (TRACK) cfl_block_store<1>(cfl, a.partials, (int64_t)blockIdx.y * gridDim.x + blockIdx.x, threadIdx.x);
}

