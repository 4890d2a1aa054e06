// Can ONE persistent launch run a chain of dependent layers on all 8 XCDs?  Protocol test + cost measurement for conv_wino's chain mode.
//   L layers x T tiles; a tile of layer l reads its own tile and 8 neighbour tiles of layer l-1 (gathered at scattered positions) and writes
//   64 KB; tiles are handed out by a ticket counter in layer-major order (a workgroup that is not resident holds no ticket, so the smallest
//   unfinished tile is always owned by a running workgroup: no deadlock whatever the residency); a tile is published by a flag after its
//   stores have completed, and consumed after the flags of its <= 9 producers were seen.  The XCDs' L2s are not coherent with each other
//   inside a kernel, so the data path is the question.  MODE:
//     0  plain loads / stores, relaxed flags            (expected to FAIL: stale L2 lines -- shows the test sees the problem)
//     1  sc1 loads / sc1 stores, relaxed agent flags    (no L2 write-back / invalidate)
//     2  plain loads / stores, release / acquire flags  (buffer_wbl2 sc1 / buffer_inv sc1 per tile)
//     3  sc1 data path, no dependencies at all          (the streaming cost alone; results not checked)
// Integer arithmetic, exact check against a host model.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_chain tools/ubench_chain.hip && tools/ubench_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TE = 16384;                                // elements (4 B) per tile: 8 rows x 32 px x 64 channels
constexpr int MAXSPIN = 1 << 16;

struct P { const unsigned* x; unsigned* b0; unsigned* b1; unsigned* ws; int L, T, tx, ty, nz; };

__device__ __forceinline__ unsigned mix(unsigned c, unsigned s, int l) { return c * 1664525u + 1013904223u + s + (unsigned)l; }

template <int MODE>
__global__ __launch_bounds__(256) void chain(P p) {
    __shared__ int sh[4];
    unsigned* ticket = p.ws; unsigned* abortw = p.ws + 1; unsigned* flags = p.ws + 64;
    const int tid = threadIdx.x;
    const long long t_start = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    for (;;) {
        if (__builtin_amdgcn_s_memrealtime() - t_start > 50000000ll) return;   // 0.5 s: never hang the box
        if (tid == 0) sh[0] = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int tk = sh[0];
        __syncthreads();
        if (tk >= p.L * p.T) return;
        const int l = tk / p.T, r = tk - l * p.T, z = r / (p.tx * p.ty), r2 = r - z * p.tx * p.ty, ty = r2 / p.tx, tx = r2 - ty * p.tx;
        int nb[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int yy = ty + i / 3 - 1, xx = tx + i % 3 - 1;
            nb[i] = (yy >= 0 && yy < p.ty && xx >= 0 && xx < p.tx) ? (z * p.ty + yy) * p.tx + xx : -1;
        }
        if (MODE != 3 && l > 0) {                        // wait for the producers (wave 0; the rest of the workgroup waits at the barrier)
            if (tid < 64) {
                const int me = tid < 9 ? nb[tid < 9 ? tid : 0] : -1;
                int ok = 0;
                for (int it = 0; it < MAXSPIN; ++it) {
                    unsigned f = 1;
                    if (tid < 9 && me >= 0) {
                        if (MODE == 2) f = __hip_atomic_load(flags + (l - 1) * p.T + me, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                        else f = __hip_atomic_load(flags + (l - 1) * p.T + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (__builtin_amdgcn_ballot_w64(f != 0) == ~0ull) { ok = 1; break; }
                    if (__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    if (__builtin_amdgcn_s_memrealtime() - t_start > 50000000ll) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                if (tid == 0) { sh[1] = ok; if (!ok) __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            }
            __syncthreads();
            if (!sh[1]) return;
        }
        const unsigned* in = l == 0 ? p.x : (l & 1) ? p.b0 : p.b1;
        unsigned* out = (l & 1) ? p.b1 : p.b0;
        __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 0x7fffffff, 0x00020000);
        const unsigned long long ob = (unsigned long long)(out + (long)r * TE);
        constexpr int AUX = (MODE == 1 || MODE == 3) ? 16 : 0;
#pragma unroll 4
        for (int q = 0; q < TE / 4 / 256; ++q) {
            const int e4 = q * 256 + tid;                // 16-byte piece of the tile
            u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(ri, (r * TE + e4 * 4) * 4, 0, AUX);
            unsigned s = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (i == 4 || nb[i] < 0) continue;
                const int pos = (e4 * 29 + i * 1031 + l) & (TE - 1);
                s += __builtin_amdgcn_raw_buffer_load_b32(ri, (nb[i] * TE + pos) * 4, 0, AUX);
            }
            u32x4 v;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = mix(c[u], s, l);
            const unsigned bo = (unsigned)e4 * 16u;
            if (MODE == 1 || MODE == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" :: "v"(bo), "v"(v), "s"(ob) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(bo), "v"(v), "s"(ob) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // FOUND HERE: with a plain `tid == 0` hipcc 7.2 threads this block into the `tid == 0` block at the top of the next iteration -- thread 0
        // takes one path round the loop, the other 63 lanes of wave 0 another, and the workgroup barriers between them are executed by wave 0
        // TWICE per iteration (once per lane set): the kernel hangs.  An opaque copy of the thread index keeps the two tests apart.
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        if (tid_o == 0 && MODE != 3) {
            if (MODE == 2) __hip_atomic_store(flags + l * p.T + r, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(flags + l * p.T + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// tickets only: is a device-scope fetch-add ONE counter for all 8 XCDs?  every ticket's cell is incremented once, the taker's XCC id recorded
__global__ __launch_bounds__(256) void tickets(unsigned* ws, int n) {
    __shared__ int sh[1];
    for (;;) {
        if (threadIdx.x == 0) sh[0] = (int)__hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int tk = sh[0];
        __syncthreads();
        if (tk >= n) return;
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ws + 64 + tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            ws[64 + n + tk] = xcc & 15u;
        }
    }
}

// the same with SCALAR atomics (s_atomic_add ... glc: returns to an SGPR, counted by lgkmcnt -- independent of the vector-memory queue a
// software-pipelined kernel keeps full): tickets by s_atomic_add, each cell incremented by a VECTOR atomic, then read back coherently by a
// scalar atomic add of 0 from every workgroup (whatever XCD it runs on) after a device-wide count says all cells are done
__device__ __forceinline__ unsigned s_fetch_add(unsigned* p, unsigned v) {
    unsigned long long a = (unsigned long long)p;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(a) : "memory");
    return v;
}
__global__ __launch_bounds__(256) void tickets_scalar(unsigned* ws, int n, unsigned* bad) {
    __shared__ int sh[1];
    for (;;) {
        int tk = 0;
        if (threadIdx.x < 64) tk = (int)s_fetch_add(ws, 1u);                 // wave 0 (uniform: one scalar instruction)
        if (threadIdx.x == 0) sh[0] = tk;
        __syncthreads();
        tk = sh[0];
        __syncthreads();
        if (tk >= n) break;
        int t_o = threadIdx.x;
        asm volatile("" : "+v"(t_o));
        if (t_o == 0) {
            __hip_atomic_fetch_add(ws + 64 + tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(ws + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // wait (bounded) until all n cells were incremented, then every workgroup reads 64 cells through scalar atomics
    if (threadIdx.x < 64) {
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        while (s_fetch_add(ws + 2, 0u) < (unsigned)n && __builtin_amdgcn_s_memrealtime() - t0 < 10000000ll) __builtin_amdgcn_s_sleep(8);
        unsigned wrong = 0;
        for (int i = 0; i < 64; ++i) wrong += s_fetch_add(ws + 64 + (blockIdx.x * 61 + i * 97) % n, 0u) != 1u;
        if (threadIdx.x == 0 && wrong) atomicAdd(bad, wrong);
    }
}

// round-trip latency of the signalling primitives under no load, one wave per workgroup on G workgroups: [0] scalar atomic add of 0 to a PRIVATE
// word, [1] the same on ONE word shared by all, [2] vector atomic (returning) on the shared word, [3] sc1 vector load of the shared word
__global__ void latency(unsigned* ws, long long* out, int iters) {
    unsigned* mine = ws + 1024 + blockIdx.x * 64;
    long long acc[4] = {};
    unsigned sink = 0;
    for (int it = 0; it < iters; ++it) {
        long long t0 = __builtin_amdgcn_s_memtime();
        sink += s_fetch_add(mine, 0u);
        long long t1 = __builtin_amdgcn_s_memtime();
        sink += s_fetch_add(ws + 512, 0u);
        long long t2 = __builtin_amdgcn_s_memtime();
        unsigned v = 0;
        if (threadIdx.x == 0) v = __hip_atomic_fetch_add(ws + 512, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sink += __builtin_amdgcn_readfirstlane(v);
        long long t3 = __builtin_amdgcn_s_memtime();
        sink += __builtin_amdgcn_readfirstlane(__hip_atomic_load(ws + 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        long long t4 = __builtin_amdgcn_s_memtime();
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3;
    }
    if (threadIdx.x == 0) { for (int i = 0; i < 4; ++i) out[blockIdx.x * 4 + i] = acc[i] / iters; ws[900] = sink; }
}

// Does vmcnt retire loads and stores in ONE order on gfx950?  [store sc1 to a cold line; load of a hot line; s_waitcnt vmcnt(1)]: with in-order
// retirement the wait ends when the STORE is acknowledged (~ the store latency), otherwise when either is (~ the load latency).
__global__ void vm_order(unsigned* cold, const unsigned* hot, long long* out, int iters) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    long long acc[3] = {};
    unsigned sink = 0;
    u4 v = {1u, 2u, 3u, threadIdx.x};
    for (int it = 0; it < iters; ++it) {
        unsigned long long cp = (unsigned long long)(cold + ((size_t)(blockIdx.x * iters + it) * 3 + 0) * 4096);
        unsigned bo = threadIdx.x * 16u;
        unsigned hv;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(bo), "v"(v), "s"(cp) : "memory");
        long long t1 = __builtin_amdgcn_s_memtime();
        cp += 16384;
        asm volatile("global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(hv) : "v"(bo & 63u), "s"((unsigned long long)hot) : "memory");
        sink += hv;
        long long t2 = __builtin_amdgcn_s_memtime();
        asm volatile("global_store_dwordx4 %1, %2, %3 sc1\n\tglobal_load_dword %0, %4, %5\n\ts_waitcnt vmcnt(1)" : "=&v"(hv) : "v"(bo), "v"(v), "s"(cp), "v"(bo & 63u), "s"((unsigned long long)hot) : "memory");
        long long t3 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sink += hv;
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2;
    }
    if (threadIdx.x == 0) { for (int i = 0; i < 3; ++i) out[blockIdx.x * 3 + i] = acc[i] / iters; cold[0] = sink; }
}

static void host_model(std::vector<unsigned>& cur, const P& p) {
    const int T = p.T;
    std::vector<unsigned> nxt(cur.size());
    for (int l = 0; l < p.L; ++l) {
        for (int r = 0; r < T; ++r) {
            const int z = r / (p.tx * p.ty), r2 = r - z * p.tx * p.ty, ty = r2 / p.tx, tx = r2 - ty * p.tx;
            int nb[9];
            for (int i = 0; i < 9; ++i) {
                const int yy = ty + i / 3 - 1, xx = tx + i % 3 - 1;
                nb[i] = (yy >= 0 && yy < p.ty && xx >= 0 && xx < p.tx) ? (z * p.ty + yy) * p.tx + xx : -1;
            }
            for (int e4 = 0; e4 < TE / 4; ++e4) {
                unsigned s = 0;
                for (int i = 0; i < 9; ++i) {
                    if (i == 4 || nb[i] < 0) continue;
                    s += cur[(size_t)nb[i] * TE + ((e4 * 29 + i * 1031 + l) & (TE - 1))];
                }
                for (int u = 0; u < 4; ++u) nxt[(size_t)r * TE + e4 * 4 + u] = cur[(size_t)r * TE + e4 * 4 + u] * 1664525u + 1013904223u + s + (unsigned)l;
            }
        }
        cur.swap(nxt);
    }
}

template <int MODE>
static int run(P p, const std::vector<unsigned>& want, int reps, const char* what) {
    const size_t n = (size_t)p.T * TE;
    std::vector<unsigned> got(n);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int bad_runs = 0;
    float best = 1e9f;
    for (int rep = 0; rep < reps; ++rep) {
        hipMemsetAsync(p.ws, 0, (64 + (size_t)p.L * p.T) * 4, 0);
        hipEventRecord(e0, 0);
        chain<MODE><<<256, 256>>>(p);
        hipEventRecord(e1, 0);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", what); return 1; }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        if (rep == 0) printf("  %s: first run %.3f ms\n", what, ms);
        unsigned ab = 0;
        hipMemcpy(&ab, p.ws + 1, 4, hipMemcpyDeviceToHost);
        hipMemcpy(got.data(), (p.L & 1) ? p.b0 : p.b1, n * 4, hipMemcpyDeviceToHost);
        size_t wrong = 0;
        if (MODE != 3) for (size_t i = 0; i < n; ++i) wrong += got[i] != want[i];
        if (wrong || ab) { ++bad_runs; if (bad_runs <= 3) printf("  %s rep %d: %zu wrong elements of %zu, abort %u\n", what, rep, wrong, n, ab); }
    }
    printf("%-52s L %d T %d: best %.3f ms = %.2f us per layer, %.2f TB/s moved; %d of %d runs wrong\n", what, p.L, p.T, best, best * 1e3 / p.L,
           (double)p.L * p.T * TE * 8 / (best * 1e-3) / 1e12, bad_runs, reps);
    return bad_runs;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    P p;
    p.L = argc > 1 ? atoi(argv[1]) : 80; p.tx = 10; p.ty = 23; p.nz = argc > 2 ? atoi(argv[2]) : 3;
    p.T = p.tx * p.ty * p.nz;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    const size_t n = (size_t)p.T * TE;
    std::vector<unsigned> x(n);
    unsigned s = 12345u;
    for (auto& v : x) { s = s * 1103515245u + 12345u; v = s; }
    unsigned *dx, *b0, *b1, *ws;
    hipMalloc(&dx, n * 4); hipMalloc(&b0, n * 4); hipMalloc(&b1, n * 4); hipMalloc(&ws, (64 + (size_t)p.L * p.T) * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    p.x = dx; p.b0 = b0; p.b1 = b1; p.ws = ws;
    std::vector<unsigned> want = x;
    host_model(want, p);
    {
        const int n = p.L * p.T;
        unsigned* w2;
        hipMalloc(&w2, (64 + 2 * (size_t)n) * 4);
        hipMemset(w2, 0, (64 + 2 * (size_t)n) * 4);
        tickets<<<256, 256>>>(w2, n);
        hipDeviceSynchronize();
        std::vector<unsigned> h(64 + 2 * (size_t)n);
        hipMemcpy(h.data(), w2, h.size() * 4, hipMemcpyDeviceToHost);
        int dup = 0, miss = 0, xh[16] = {};
        for (int i = 0; i < n; ++i) { dup += h[64 + i] > 1; miss += h[64 + i] == 0; xh[h[64 + n + i] & 15]++; }
        printf("tickets: counter ended at %u (want %d + 256), %d cells taken twice, %d never; by XCC:", h[0], n, dup, miss);
        for (int i = 0; i < 8; ++i) printf(" %d", xh[i]);
        printf("\n");
    }
    {
        const int n = p.L * p.T;
        unsigned* w2;
        hipMalloc(&w2, (64 + 2 * (size_t)n) * 4);
        hipMemset(w2, 0, (64 + 2 * (size_t)n) * 4);
        tickets_scalar<<<256, 256>>>(w2, n, w2 + 3);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned> h(64 + (size_t)n);
        hipMemcpy(h.data(), w2, h.size() * 4, hipMemcpyDeviceToHost);
        int dup = 0, miss = 0;
        for (int i = 0; i < n; ++i) { dup += h[64 + i] > 1; miss += h[64 + i] == 0; }
        printf("scalar tickets (%s): counter ended at %u (want %d + 256), %d cells taken twice, %d never; scalar read-back saw %u stale cells\n",
               hipGetErrorString(e), h[0], n, dup, miss, h[3]);
    }
    for (int G : {1, 256}) {
        unsigned* w3; long long* o3;
        hipMalloc(&w3, (1024 + 256 * 64) * 4); hipMalloc(&o3, 256 * 4 * 8);
        hipMemset(w3, 0, (1024 + 256 * 64) * 4);
        latency<<<G, 64>>>(w3, o3, 200);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * 4);
        hipMemcpy(h.data(), o3, G * 4 * 8, hipMemcpyDeviceToHost);
        double m[4] = {};
        for (int b = 0; b < G; ++b) for (int i = 0; i < 4; ++i) m[i] += (double)h[b * 4 + i] / G;
        printf("latency, %3d workgroups (shader cycles): scalar atomic private %.0f | scalar atomic shared %.0f | vector atomic shared %.0f | sc1 load %.0f\n", G, m[0], m[1], m[2], m[3]);
    }
    {
        unsigned *cold, *hot; long long* o4;
        const int G4 = 64, it4 = 64;
        hipMalloc(&cold, (size_t)G4 * it4 * 3 * 4096 * 4 + 65536); hipMalloc(&hot, 4096); hipMalloc(&o4, G4 * 3 * 8);
        hipMemset(hot, 0, 4096);
        vm_order<<<G4, 64>>>(cold, hot, o4, it4);
        hipDeviceSynchronize();
        std::vector<long long> h(G4 * 3);
        hipMemcpy(h.data(), o4, G4 * 3 * 8, hipMemcpyDeviceToHost);
        double m[3] = {};
        for (int b = 0; b < G4; ++b) for (int i = 0; i < 3; ++i) m[i] += (double)h[b * 3 + i] / G4;
        printf("vmcnt order (cycles): sc1 store alone %.0f | hot load alone %.0f | [store; load; vmcnt(1)] %.0f  -> %s\n", m[0], m[1], m[2],
               m[2] > 0.7 * m[0] ? "in order (the wait covers the store)" : "OUT of order (the wait ended with the load)");
    }
    int bad = 0;
    run<3>(p, want, 3, "3: sc1 data path, no dependencies (not checked)");
    bad += run<1>(p, want, reps, "1: sc1 loads + sc1 stores, relaxed agent flags");
    bad += run<2>(p, want, reps, "2: plain data path, release / acquire agent flags");
    run<0>(p, want, reps, "0: plain everything (expected to fail)");
    printf(bad ? "PROTOCOL FAILED\n" : "modes 1 and 2 exact\n");
    return bad != 0;
}
