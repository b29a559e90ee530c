// tests/host_emul/coop_wave.h -- TEST INFRASTRUCTURE.  A workgroup of lock-step lanes on the host, for the per-wave kernels of
// curve25519_amd/csrc/coop25519.cuh / coop_ops.cuh: every lane is a fiber (ucontext) of ONE host thread, and the cross-lane
// primitives those headers use -- DPP row moves, v_permlane16_swap / v_permlane32_swap, the wave barrier in front of and
// behind LDS traffic between lanes, __syncthreads between the waves of a workgroup -- are rendezvous points: a lane that
// reaches one publishes its value and yields until all lanes of its wave (workgroup) have arrived.  Between two rendezvous a
// lane runs alone, so an exchange through LDS that the device source does NOT bracket with wave_fence() shows up here as a
// wrong result (on the device the lanes of a wave execute in lock-step, but the fence is also what keeps the compiler from
// moving the accesses).  The model of the instructions is the ISA's: quad_perm within groups of four lanes (quad25519.cuh),
// row_shr:n / row_ror:n within 16-lane rows with
// bound_ctrl (zero where nothing arrives), permlane16_swap = odd rows of the first operand <-> even rows of the second,
// permlane32_swap = upper half of the first <-> lower half of the second.  The GPU suite establishes that the hardware
// agrees; this establishes the arithmetic and data movement of the cooperative formulas on a CPU-only machine.
// Included by valu_model.h (declarations) and emul.cpp (EMUL_COOP_WAVE_IMPL: the scheduler).
#ifndef EMUL_COOP_WAVE_DECLARED
#define EMUL_COOP_WAVE_DECLARED
#include <stdint.h>

namespace emul_coop {

struct pair { uint32_t v[2]; uint32_t operator[](int i) const { return v[i]; } };
uint32_t dpp(uint32_t old, uint32_t src, int ctrl, bool bound_ctrl);
pair swap16(uint32_t a, uint32_t b);
pair swap32(uint32_t a, uint32_t b);
void wave_sync();                       // __builtin_amdgcn_wave_barrier (a no-op outside run_block)
void block_sync();                      // __syncthreads (likewise)
unsigned long long sync_points();       // rendezvous counted since the process started (the tests report it)

}  // namespace emul_coop
#endif

#if defined(EMUL_COOP_WAVE_IMPL) && !defined(EMUL_COOP_WAVE_DEFINED)
#define EMUL_COOP_WAVE_DEFINED
#include <ucontext.h>

#include <functional>
#include <stdexcept>
#include <vector>

namespace emul_coop {

constexpr int MAX_THREADS = 256, STACK_BYTES = 512 << 10;

struct Block {
    int threads = 0, cur = -1, live = 0;
    ucontext_t main_ctx;
    ucontext_t ctx[MAX_THREADS];
    bool done[MAX_THREADS];
    uint32_t xchg[MAX_THREADS], xchg2[MAX_THREADS];
    unsigned wave_gen[MAX_THREADS / 64], wave_arrived[MAX_THREADS / 64], wave_live[MAX_THREADS / 64];
    unsigned blk_gen = 0, blk_arrived = 0;
    const std::function<void()>* body = nullptr;
};
static thread_local Block* g_block = nullptr;
static std::vector<char> g_stacks;      // reused between blocks (one emulated workgroup at a time per process: run_block's lock)
static unsigned long long g_sync_points = 0;

unsigned long long sync_points() { return g_sync_points; }

// hand the host thread to the next lane that has not finished (round robin), or back to run_block when none is left
static void yield_from(Block* b, int me)
{
    int next = -1;
    for (int i = 1; i <= b->threads; i++) {
        const int c = (me + i) % b->threads;
        if (!b->done[c]) { next = c; break; }
    }
    if (next == me) return;
    ucontext_t* from = me >= 0 ? &b->ctx[me] : &b->main_ctx;
    ucontext_t* to = next >= 0 ? &b->ctx[next] : &b->main_ctx;
    b->cur = next;
    emul_tid.x = next >= 0 ? (unsigned)next : 0u;
    swapcontext(from, to);
    b->cur = me;                          // resumed
    emul_tid.x = me >= 0 ? (unsigned)me : 0u;
}

static void rendezvous(unsigned& gen, unsigned& arrived, const unsigned& live)
{
    Block* b = g_block;
    const int me = b->cur;
    g_sync_points++;
    const unsigned g = gen;
    if (++arrived >= live) { arrived = 0; gen++; return; }
    while (gen == g) {
        yield_from(b, me);
        if (gen == g && b->live == 1) throw std::logic_error("emul_coop: a lane waits at a rendezvous nobody else can reach");
    }
}

void wave_sync()
{
    Block* b = g_block;
    if (!b) return;
    const int w = b->cur >> 6;
    rendezvous(b->wave_gen[w], b->wave_arrived[w], b->wave_live[w]);
}

void block_sync()
{
    Block* b = g_block;
    if (!b) return;
    static thread_local unsigned live;
    live = (unsigned)b->live;
    rendezvous(b->blk_gen, b->blk_arrived, live);
}

uint32_t dpp(uint32_t old, uint32_t src, int ctrl, bool bound_ctrl)
{
    Block* b = g_block;
    if (!b) throw std::logic_error("emul_coop: a DPP move outside run_block");
    const int me = b->cur, row = me & ~15, c = me & 15;
    b->xchg[me] = src;
    wave_sync();
    int from = -1;
    if (ctrl >= 0 && ctrl <= 0xff) from = (me & ~3) + ((ctrl >> (2 * (me & 3))) & 3);                                  // quad_perm:[a,b,c,d]
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; if (c >= n) from = row + c - n; }              // row_shr:n
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl - 0x100; if (c + n < 16) from = row + c + n; }      // row_shl:n
    else if (ctrl >= 0x121 && ctrl <= 0x12f) { const int n = ctrl - 0x120; from = row + ((c - n) & 15); }             // row_ror:n
    else throw std::logic_error("emul_coop: DPP control not modelled");
    const uint32_t r = from >= 0 ? b->xchg[from] : (bound_ctrl ? 0u : old);
    wave_sync();
    return r;
}

// V_PERMLANE16_SWAP: in each 32-lane half, the odd row of vdst (first operand) and the even row of src0 (second) change places
pair swap16(uint32_t a, uint32_t bb)
{
    Block* b = g_block;
    if (!b) throw std::logic_error("emul_coop: v_permlane16_swap outside run_block");
    const int me = b->cur;
    b->xchg[me] = a;
    b->xchg2[me] = bb;
    wave_sync();
    pair r;
    const bool odd_row = (me >> 4) & 1;
    r.v[0] = odd_row ? b->xchg2[me - 16] : a;      // vdst: its odd row now holds src0's even row
    r.v[1] = odd_row ? bb : b->xchg[me + 16];      // src0: its even row now holds vdst's odd row
    wave_sync();
    return r;
}

// V_PERMLANE32_SWAP: the upper 32 lanes of vdst and the lower 32 lanes of src0 change places
pair swap32(uint32_t a, uint32_t bb)
{
    Block* b = g_block;
    if (!b) throw std::logic_error("emul_coop: v_permlane32_swap outside run_block");
    const int me = b->cur;
    b->xchg[me] = a;
    b->xchg2[me] = bb;
    wave_sync();
    pair r;
    const bool upper = (me >> 5) & 1;
    r.v[0] = upper ? b->xchg2[me - 32] : a;
    r.v[1] = upper ? bb : b->xchg[me + 32];
    wave_sync();
    return r;
}

static void lane_entry()
{
    Block* b = g_block;
    const int me = b->cur;
    emul_tid.x = (unsigned)me;
    (*b->body)();
    // this lane is done: the rendezvous the others wait at no longer count it
    b->done[me] = true;
    b->live--;
    const int w = me >> 6;
    b->wave_live[w]--;
    if (b->wave_live[w] && b->wave_arrived[w] >= b->wave_live[w]) { b->wave_arrived[w] = 0; b->wave_gen[w]++; }
    if (b->live && b->blk_arrived >= (unsigned)b->live) { b->blk_arrived = 0; b->blk_gen++; }
    yield_from(b, me);                    // never comes back
    throw std::logic_error("emul_coop: a finished lane was resumed");
}

// run `body` once per lane of a workgroup of `threads` lanes (threadIdx.x = the lane), lock-step at the rendezvous points
inline void run_block(int threads, const std::function<void()>& body)
{
    if (threads < 1 || threads > MAX_THREADS || g_block) throw std::logic_error("emul_coop: bad workgroup");
    static Block blk;
    Block* b = &blk;
    if (g_stacks.size() < (size_t)threads * STACK_BYTES) g_stacks.resize((size_t)threads * STACK_BYTES);
    b->threads = threads;
    b->live = threads;
    b->body = &body;
    b->blk_gen = b->blk_arrived = 0;
    for (int w = 0; w < MAX_THREADS / 64; w++) {
        b->wave_gen[w] = b->wave_arrived[w] = 0;
        const int in_wave = threads - 64 * w;
        b->wave_live[w] = in_wave <= 0 ? 0u : in_wave > 64 ? 64u : (unsigned)in_wave;
    }
    for (int i = 0; i < threads; i++) {
        b->done[i] = false;
        getcontext(&b->ctx[i]);
        b->ctx[i].uc_stack.ss_sp = g_stacks.data() + (size_t)i * STACK_BYTES;
        b->ctx[i].uc_stack.ss_size = STACK_BYTES;
        b->ctx[i].uc_link = nullptr;
        makecontext(&b->ctx[i], lane_entry, 0);
    }
    g_block = b;
    b->cur = -1;
    const emul_dim3 saved = emul_tid;
    yield_from(b, -1);                    // lane 0 first; comes back when every lane has finished
    g_block = nullptr;
    emul_tid = saved;
}

}  // namespace emul_coop
#endif
