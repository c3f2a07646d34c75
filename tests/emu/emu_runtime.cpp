// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE: cooperative-fiber runtime behind
// tests/emu/lina_dev.h.  One workgroup at a time; every GPU thread is a ucontext fiber
// scheduled round-robin; __syncthreads and wave collectives are yield points.  A pass of the
// scheduler that makes no progress means a divergent barrier -> abort with a message.
#include <lina_dev.h>
#include <ucontext.h>
#include <vector>

namespace lina_emu {

struct Fiber {
    ucontext_t ctx;
    dim3 tid;
    int linear = 0;
    bool done = false;
    unsigned wave_parity = 0;
    char* stack = nullptr;
    std::vector<std::pair<void*, const void*>> pending;   // LINA_EMU_DMA_LATE: issued, not yet landed
};

struct Wave {
    int count = 0;
    unsigned gen = 0;
    uint32_t buf[2][64 * 16];
};

Fiber* cur = nullptr;
dim3 g_blockIdx, g_blockDim, g_gridDim;
unsigned char* g_dyn_smem = nullptr;

static ucontext_t sched_ctx;
static std::vector<Fiber> fibers;
static std::vector<Wave> waves;
static int bar_count = 0;
static unsigned bar_gen = 0;
static unsigned long progress = 0;
static const std::function<void()>* g_body = nullptr;
static const size_t kStack = 256 * 1024;

const dim3& cur_tid() { return cur->tid; }
int cur_lane() { return cur->linear & 63; }

bool dma_late() {
    static const bool late = getenv("LINA_EMU_DMA_LATE") && atoi(getenv("LINA_EMU_DMA_LATE")) != 0;
    return late;
}
void dma_defer(void* dst, const void* src) { cur->pending.emplace_back(dst, src); }
void dma_flush_mine() {
    for (auto& p : cur->pending) memcpy(p.first, p.second, 16);
    cur->pending.clear();
}

static void yield() { swapcontext(&cur->ctx, &sched_ctx); }

void syncthreads() {
    const unsigned g = bar_gen;
    if (++bar_count == (int)fibers.size()) {
        bar_count = 0;
        ++bar_gen;
        ++progress;
    } else {
        while (bar_gen == g) yield();
    }
}

void wave_exchange(const uint32_t* mine, int n, uint32_t* out) {
    Wave& w = waves[cur->linear >> 6];
    const unsigned p = cur->wave_parity & 1;
    cur->wave_parity++;
    const int lane = cur->linear & 63;
    for (int i = 0; i < n; ++i) w.buf[p][lane * n + i] = mine[i];
    const unsigned g = w.gen;
    if (++w.count == 64) {
        w.count = 0;
        ++w.gen;
        ++progress;
    } else {
        while (w.gen == g) yield();
    }
    memcpy(out, w.buf[p], sizeof(uint32_t) * 64 * n);
}

static void trampoline() {
    (*g_body)();
    dma_flush_mine();
    cur->done = true;
    ++progress;
    swapcontext(&cur->ctx, &sched_ctx);
}

void launch_impl(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem) {
    const int nthr = (int)(block.x * block.y * block.z);
    if (nthr % 64 != 0) {
        fprintf(stderr, "lina_emu: block size %d is not a multiple of the wave size 64\n", nthr);
        abort();
    }
    g_body = &body;
    g_gridDim = grid;
    g_blockDim = block;
    std::vector<unsigned char> dyn(smem + 64);
    g_dyn_smem = (unsigned char*)(((uintptr_t)dyn.data() + 15) & ~(uintptr_t)15);
    static std::vector<char*> stacks;
    while ((int)stacks.size() < nthr) stacks.push_back((char*)malloc(kStack));
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = dim3(bx, by, bz);
                fibers.assign(nthr, Fiber());
                waves.assign(nthr / 64, Wave());
                bar_count = 0;
                for (int t = 0; t < nthr; ++t) {
                    Fiber& f = fibers[t];
                    f.linear = t;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.stack = stacks[t];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &sched_ctx;
                    makecontext(&f.ctx, trampoline, 0);
                }
                // LINA_EMU_SHUFFLE=seed: the waves are visited in a different pseudo-random order on every scheduler pass (lanes of
                // a wave stay in order), so that a kernel whose result depends on which wave gets ahead between two barriers
                // (an LDS race) does not pass on the one fixed interleaving
                static const unsigned shuffle_seed = getenv("LINA_EMU_SHUFFLE") ? (unsigned)atoi(getenv("LINA_EMU_SHUFFLE")) : 0u;
                unsigned rng = shuffle_seed * 2654435761u + bx * 40503u + 12345u;
                const int nwaves = nthr / 64;
                std::vector<int> worder(nwaves);
                for (int i = 0; i < nwaves; ++i) worder[i] = i;
                int live = nthr;
                while (live > 0) {
                    const unsigned long before = progress;
                    live = 0;
                    if (shuffle_seed)
                        for (int i = nwaves - 1; i > 0; --i) {
                            rng = rng * 1664525u + 1013904223u;
                            std::swap(worder[i], worder[(rng >> 8) % (unsigned)(i + 1)]);
                        }
                    for (int tt = 0; tt < nthr; ++tt) {
                        const int t = worder[tt >> 6] * 64 + (tt & 63);
                        if (fibers[t].done) continue;
                        cur = &fibers[t];
                        swapcontext(&sched_ctx, &fibers[t].ctx);
                        if (!fibers[t].done) ++live;
                    }
                    if (live > 0 && progress == before) {
                        fprintf(stderr, "lina_emu: deadlock (divergent barrier / wave collective) in block (%u,%u,%u)\n",
                                bx, by, bz);
                        abort();
                    }
                }
            }
    cur = nullptr;
    g_dyn_smem = nullptr;
}

}  // namespace lina_emu
