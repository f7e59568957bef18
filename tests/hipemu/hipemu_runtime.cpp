// TEST-ONLY: fiber scheduler behind tests/hipemu/hip/hip_runtime.h (see that header).
#include "hip/hip_runtime.h"

#include <sys/mman.h>
#include <vector>

extern "C" void hipemu_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {

static const size_t STACK_BYTES = 512 * 1024;
static const int MAX_THREADS = 1024;

struct Wave {
    int alive = 0;
    int arrived = 0;
    unsigned gen = 0;
    uint32_t slot[2][64];
    float fa[2][64], fb[2][64];
    float fa8[2][64][8], fb8[2][64][8];
    uint64_t ballot[2];
};

struct Lane {
    LaneView view;
    void* sp = nullptr;
    bool done = false;
    int wave = 0;
    int lane = 0;
};

static Lane g_lanes[MAX_THREADS];
static Wave g_waves[MAX_THREADS / 64];
static char* g_stacks = nullptr;
static void* g_main_sp = nullptr;
static int g_cur = -1;
static int g_nthreads = 0;
static int g_alive = 0;
static int g_bar_arrived = 0;
static unsigned g_bar_gen = 0;
static unsigned long g_progress = 0;
static const std::function<void()>* g_body = nullptr;
LaneView* cur_view = nullptr;

static void yield_to_main() {
    Lane& l = g_lanes[g_cur];
    hipemu_switch(&l.sp, g_main_sp);
}

static void lane_exit() {
    Lane& l = g_lanes[g_cur];
    l.done = true;
    g_alive--;
    Wave& w = g_waves[l.wave];
    w.alive--;
    g_progress++;
    // release barriers that were only waiting for this lane
    if (g_alive > 0 && g_bar_arrived >= g_alive) { g_bar_arrived = 0; g_bar_gen++; }
    if (w.alive > 0 && w.arrived >= w.alive) { w.arrived = 0; w.gen++; }
    for (;;) yield_to_main();
}

extern "C" void hipemu_trampoline() {
    (*g_body)();
    lane_exit();
}

// a lane that polls memory (software barrier): let the other lanes run; no progress of its own (a block whose live lanes all spin is a deadlock)
void spin_yield() { yield_to_main(); }

void syncthreads() {
    unsigned g = g_bar_gen;
    g_bar_arrived++;
    g_progress++;
    if (g_bar_arrived >= g_alive) { g_bar_arrived = 0; g_bar_gen++; return; }
    while (g_bar_gen == g) yield_to_main();
}

static inline void wave_barrier(Wave& w) {
    unsigned g = w.gen;
    w.arrived++;
    g_progress++;
    if (w.arrived >= w.alive) { w.arrived = 0; w.gen++; return; }
    while (w.gen == g) yield_to_main();
}

int lane_id() { return g_lanes[g_cur].lane; }

uint32_t wave_exchange_u32(uint32_t v, int src_lane) {
    Lane& l = g_lanes[g_cur];
    Wave& w = g_waves[l.wave];
    int b = w.gen & 1;
    w.slot[b][l.lane] = v;
    wave_barrier(w);
    return w.slot[b][src_lane & 63];
}

uint64_t wave_ballot(bool p) {
    Lane& l = g_lanes[g_cur];
    Wave& w = g_waves[l.wave];
    int b = w.gen & 1;
    if (w.arrived == 0) w.ballot[b] = 0;
    if (p) w.ballot[b] |= (1ull << l.lane);
    wave_barrier(w);
    return w.ballot[b];
}

void wave_mfma32x32x2(float a, float b, const float* c, float* d) {
    Lane& l = g_lanes[g_cur];
    Wave& w = g_waves[l.wave];
    if (w.alive != 64) { fprintf(stderr, "hipemu: MFMA with %d live lanes\n", w.alive); abort(); }
    int bf = w.gen & 1;
    w.fa[bf][l.lane] = a;
    w.fb[bf][l.lane] = b;
    wave_barrier(w);
    int j = l.lane & 31, h = l.lane >> 5;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = std::fmaf(w.fa[bf][i], w.fb[bf][j], c[r]);
        d[r] = std::fmaf(w.fa[bf][i + 32], w.fb[bf][j + 32], acc);
    }
}

void wave_mfma32x32x16_bf16(const float* a8, const float* b8, const float* c, float* d) {
    // A: lane l holds A[i=l&31][k=8*(l>>5)+e]; B: B[k=8*(l>>5)+e][j=l&31]; e = 0..7 (bf16 values widened by the caller)
    Lane& l = g_lanes[g_cur];
    Wave& w = g_waves[l.wave];
    if (w.alive != 64) { fprintf(stderr, "hipemu: MFMA with %d live lanes\n", w.alive); abort(); }
    int bf = w.gen & 1;
    for (int e = 0; e < 8; ++e) { w.fa8[bf][l.lane][e] = a8[e]; w.fb8[bf][l.lane][e] = b8[e]; }
    wave_barrier(w);
    int j = l.lane & 31, h = l.lane >> 5;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        for (int hh = 0; hh < 2; ++hh)
            for (int e = 0; e < 8; ++e) acc = std::fmaf(w.fa8[bf][i + 32 * hh][e], w.fb8[bf][j + 32 * hh][e], acc);
        d[r] = acc;
    }
}

void wave_mfma16x16x32(const float* a8, const float* b8, const float* c, float* d) {
    // v_mfma_f32_16x16x32_{f16,bf16}: A: lane l holds A[i=l&15][k=8*(l>>4)+e]; B: B[k=8*(l>>4)+e][j=l&15]; C/D: col=l&15, row=(l>>4)*4+r
    Lane& l = g_lanes[g_cur];
    Wave& w = g_waves[l.wave];
    if (w.alive != 64) { fprintf(stderr, "hipemu: MFMA with %d live lanes\n", w.alive); abort(); }
    int bf = w.gen & 1;
    for (int e = 0; e < 8; ++e) { w.fa8[bf][l.lane][e] = a8[e]; w.fb8[bf][l.lane][e] = b8[e]; }
    wave_barrier(w);
    int j = l.lane & 15, q = l.lane >> 4;
    for (int r = 0; r < 4; ++r) {
        int i = q * 4 + r;
        float acc = c[r];
        for (int kq = 0; kq < 4; ++kq)
            for (int e = 0; e < 8; ++e) acc = std::fmaf(w.fa8[bf][i + 16 * kq][e], w.fb8[bf][j + 16 * kq][e], acc);
        d[r] = acc;
    }
}

void wave_mfma16x16x4(float a, float b, const float* c, float* d) {
    // A: lane l holds A[i=l&15][k=l>>4]; B: B[k=l>>4][j=l&15]; C/D: col=l&15, row=(l>>4)*4+r
    Lane& l = g_lanes[g_cur];
    Wave& w = g_waves[l.wave];
    if (w.alive != 64) { fprintf(stderr, "hipemu: MFMA with %d live lanes\n", w.alive); abort(); }
    int bf = w.gen & 1;
    w.fa[bf][l.lane] = a;
    w.fb[bf][l.lane] = b;
    wave_barrier(w);
    int j = l.lane & 15, q = l.lane >> 4;
    for (int r = 0; r < 4; ++r) {
        int i = q * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fmaf(w.fa[bf][i + 16 * k], w.fb[bf][j + 16 * k], acc);
        d[r] = acc;
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || nt > MAX_THREADS) { fprintf(stderr, "hipemu: bad block size %d\n", nt); abort(); }
    if (g_cur != -1) { fprintf(stderr, "hipemu: nested launch\n"); abort(); }
    if (!g_stacks) {
        g_stacks = (char*)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == MAP_FAILED) { perror("hipemu mmap"); abort(); }
    }
    g_body = &body;
    g_nthreads = nt;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        int nwaves = (nt + 63) / 64;
        for (int wv = 0; wv < nwaves; ++wv) { g_waves[wv] = Wave(); }
        for (int t = 0; t < nt; ++t) {
            Lane& l = g_lanes[t];
            l.view.tid = uint3{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
            l.view.bid = uint3{bx, by, bz};
            l.view.bdim = block;
            l.view.gdim = grid;
            l.done = false;
            l.wave = t / 64;
            l.lane = t % 64;
            g_waves[l.wave].alive++;
            char* top = g_stacks + STACK_BYTES * (size_t)(t + 1);
            void** sp = (void**)(top - 64);
            for (int q = 0; q < 6; ++q) sp[q] = nullptr;
            sp[6] = (void*)&hipemu_trampoline;
            sp[7] = nullptr;
            l.sp = (void*)sp;
        }
        g_alive = nt;
        g_bar_arrived = 0;
        g_bar_gen = 0;
        while (g_alive > 0) {
            unsigned long before = g_progress;
            for (int t = 0; t < nt; ++t) {
                if (g_lanes[t].done) continue;
                g_cur = t;
                cur_view = &g_lanes[t].view;
                hipemu_switch(&g_main_sp, g_lanes[t].sp);
            }
            if (g_alive > 0 && g_progress == before) {
                fprintf(stderr, "hipemu: deadlock (divergent barrier / wave collective) in block (%u,%u,%u); "
                        "%d threads alive, %d at block barrier\n", bx, by, bz, g_alive, g_bar_arrived);
                abort();
            }
        }
    }
    g_cur = -1;
    cur_view = nullptr;
    g_body = nullptr;
}

}  // namespace hipemu

// ---- device allocations with a canary behind the requested size: a kernel writing past its buffer aborts the test run at
// hipFree with the size of the offending allocation instead of corrupting the host heap
#include <map>
namespace hipemu {
static std::map<void*, size_t>& alloc_map() { static std::map<void*, size_t> m; return m; }
static const size_t GUARD = 1024;
void* guarded_alloc(size_t n) {
    const size_t tot = (n + GUARD + 255) / 256 * 256 + 256;
    char* p = (char*)std::aligned_alloc(256, tot);
    if (!p) return nullptr;
    std::memset(p + n, 0xA5, GUARD);
    alloc_map()[p] = n;
    return p;
}
void guarded_free(void* p) {
    if (!p) return;
    auto it = alloc_map().find(p);
    if (it != alloc_map().end()) {
        const unsigned char* g = (const unsigned char*)p + it->second;
        for (size_t i = 0; i < GUARD; ++i)
            if (g[i] != 0xA5) { std::fprintf(stderr, "hipemu: write past the end of a %zu-byte device allocation (offset +%zu)\n", it->second, i); std::abort(); }
        alloc_map().erase(it);
    }
    std::free(p);
}
}  // namespace hipemu
