// pair_auto.hip -- plan cache + change detector behind the HOOMD-signature entry points
// (see pair_auto.hpp for the protocol).
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "azp_device.hpp"
#include "pair_auto.hpp"

namespace azp
{
namespace
{
constexpr uint64_t AUTO_FULL_HASH_ENTRIES = 1ull << 22; // lists up to this size are fingerprinted entry by entry
constexpr size_t AUTO_MAX_PLANS = 8;                    // cache capacity (least recently used plan is evicted)

struct AutoState // device words of one cached plan (pinned host mirror: the first 48 bytes)
    {
    TileDyn dyn;                        // what the speculatively launched tile kernel reads
    unsigned long long fingerprint;     // of this call
    unsigned long long d2_bits;         // largest |dx|^2 since the compile
    uint32_t types_changed;
    uint32_t _pad;
    // device only
    unsigned long long expected[8];     // fingerprint learned at the compile, per sample phase
    };
constexpr size_t AUTO_STATE_HOST_BYTES = 48;

struct CheckKArgs
    {
    const double* pos;
    const double* pos0;       // positions when the plan was compiled (null: no plan yet)
    const uint32_t* n_neigh;
    const uint64_t* head_list;
    const uint32_t* nlist;
    const double* rcutsq;
    unsigned long long* out;  // per workgroup: fingerprint part (sum of mixed words), max |dx|^2 bits, type change flag
    AutoState* state;
    BoxDev box;
    double shell_winv;        // of the cached plan (0: no shells -> whole rows unless nothing moved)
    unsigned long long generation; // caller's list generation (0: none -> the fingerprint decides)
    uint32_t N, n_max, ntypes;
    uint32_t full;            // 1: every list entry enters the fingerprint
    uint32_t phase;           // sampled fingerprints: rows with (i & 7) == phase contribute two entries
    uint32_t learn;           // 1: store the fingerprint as expected[phase] (right after a compile)
    uint32_t list_words;      // 0: the caller vouches for the list (generation): fingerprint of N, box, cutoffs only
    };

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
    {
    // SplitMix64 finaliser: the fingerprint is the wrapping SUM of mixed (position, value)
    // words, so the order in which threads add does not matter
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
    }

// Grid-stride over the particles; every workgroup leaves ONE partial triple in out[3 * block ..];
// auto_fold_kernel folds them.
constexpr uint32_t AUTO_CHECK_BLOCKS = 1024;

__global__ void __launch_bounds__(256) auto_check_kernel(const CheckKArgs a)
    {
    unsigned long long h = 0;
    double d2 = 0.0;
    uint32_t type_changed = 0;
    const uint32_t n_items = max(max(a.N, a.n_max), a.ntypes * a.ntypes);
    // U items per thread and trip, every load of a trip issued before the first use (the kernel is a stream of
    // 80 B per particle; with one item in flight per thread it ran at a third of the memory rate)
    constexpr int U = 4;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i0 = blockIdx.x * 256u + threadIdx.x; i0 < n_items; i0 += U * stride)
        {
        uint32_t n[U];
        uint64_t head[U];
        double4 p[U], q[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            {
            const uint32_t i = i0 + (uint32_t)u * stride;
            const bool row = i < a.N && a.list_words;
            n[u] = row ? a.n_neigh[i] : 0u;
            head[u] = row ? a.head_list[i] : 0ull;
            if (a.pos0 && i < a.n_max)
                {
                p[u] = load_scalar4(a.pos, i);
                q[u] = load_scalar4(a.pos0, i);
                }
            else
                p[u] = q[u] = make_double4(0.0, 0.0, 0.0, 0.0);
            }
        uint32_t last[U], mid[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            {
            const uint32_t i = i0 + (uint32_t)u * stride;
            const bool sampled = i < a.N && a.list_words && !a.full && (i & 7u) == a.phase && n[u];
            const uint32_t* row = a.nlist + head[u];
            last[u] = sampled ? row[n[u] - 1] : 0u;
            mid[u] = sampled ? row[n[u] >> 1] : 0u;
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
            {
            const uint32_t i = i0 + (uint32_t)u * stride;
            if (i < a.N && a.list_words)
                {
                h += mix64(((unsigned long long)i << 32) ^ n[u]) + mix64(head[u] * 0x100000001B3ull + i);
                if (a.full)
                    {
                    const uint32_t* row = a.nlist + head[u];
                    for (uint32_t k = 0; k < n[u]; ++k)
                        h += mix64(((head[u] + k) << 32) ^ row[k] ^ 0xA5A5A5A500000000ull);
                    }
                else if ((i & 7u) == a.phase && n[u])
                    h += mix64(((head[u] + n[u] - 1) << 32) ^ last[u]) + mix64(((head[u] + (n[u] >> 1)) << 32) ^ mid[u]);
                }
            if (i < a.ntypes * a.ntypes)
                h += mix64(__double_as_longlong(a.rcutsq[i]) + 0x1234567ull * (i + 1));
            if (i == 0)
                {
                h += mix64(__double_as_longlong(a.box.Lx)) + mix64(__double_as_longlong(a.box.Ly) + 1) + mix64(__double_as_longlong(a.box.Lz) + 2)
                     + mix64(__double_as_longlong(a.box.xy) + 3) + mix64(__double_as_longlong(a.box.xz) + 4)
                     + mix64(__double_as_longlong(a.box.yz) + 5) + mix64(((unsigned long long)a.N << 32) | a.n_max) + mix64(0x77ull + a.ntypes)
                     + mix64(a.generation ^ 0x5bd1e995ull);
                }
            if (a.pos0 && i < a.n_max)
                {
                double dx = p[u].x - q[u].x, dy = p[u].y - q[u].y, dz = p[u].z - q[u].z;
                min_image(a.box, dx, dy, dz);
                double d = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
                if (!(d == d))
                    d = 1.0e300; // NaN position: whole rows
                d2 = fmax(d2, d);
                type_changed |= (type_from_w(p[u].w) != type_from_w(q[u].w)) ? 1u : 0u;
                }
            }
        }
    for (int off = 32; off > 0; off >>= 1)
        {
        h += (unsigned long long)__shfl_xor((long long)h, off, 64);
        d2 = fmax(d2, __shfl_xor(d2, off, 64));
        type_changed |= (uint32_t)__shfl_xor((int)type_changed, off, 64);
        }
    __shared__ unsigned long long s_h[4];
    __shared__ double s_d[4];
    __shared__ uint32_t s_t[4];
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0)
        {
        s_h[wave] = h; s_d[wave] = d2; s_t[wave] = type_changed;
        }
    __syncthreads();
    if (threadIdx.x == 0)
        {
        a.out[3 * blockIdx.x] = s_h[0] + s_h[1] + s_h[2] + s_h[3];
        a.out[3 * blockIdx.x + 1] = (unsigned long long)__double_as_longlong(fmax(fmax(s_d[0], s_d[1]), fmax(s_d[2], s_d[3])));
        a.out[3 * blockIdx.x + 2] = s_t[0] | s_t[1] | s_t[2] | s_t[3];
        }
    }

// One workgroup folds the partials of the check into the state words the tile kernel and the host read. A kernel
// of its own: "the last workgroup folds" needs a device-scope fence and a same-address atomic per workgroup, and
// across the eight XCDs those cost ~60 ns each, one after the other (1,024 workgroups: 65 us; measured).
__global__ void __launch_bounds__(256) auto_fold_kernel(const CheckKArgs a, const uint32_t n_blocks)
    {
    unsigned long long h = 0;
    double d2 = 0.0;
    uint32_t type_changed = 0;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 256u)
        {
        h += a.out[3 * b];
        d2 = fmax(d2, __longlong_as_double((long long)a.out[3 * b + 1]));
        type_changed |= (uint32_t)a.out[3 * b + 2];
        }
    for (int off = 32; off > 0; off >>= 1)
        {
        h += (unsigned long long)__shfl_xor((long long)h, off, 64);
        d2 = fmax(d2, __shfl_xor(d2, off, 64));
        type_changed |= (uint32_t)__shfl_xor((int)type_changed, off, 64);
        }
    __shared__ unsigned long long s_h[4];
    __shared__ double s_d[4];
    __shared__ uint32_t s_t[4];
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0)
        {
        s_h[wave] = h; s_d[wave] = d2; s_t[wave] = type_changed;
        }
    __syncthreads();
    if (threadIdx.x == 0)
        {
        AutoState& st = *a.state;
        const unsigned long long fp = s_h[0] + s_h[1] + s_h[2] + s_h[3];
        const double dd = fmax(fmax(s_d[0], s_d[1]), fmax(s_d[2], s_d[3]));
        const uint32_t tc = s_t[0] | s_t[1] | s_t[2] | s_t[3];
        if (a.learn)
            st.expected[a.phase] = fp;
        const bool stale = !a.learn && (!a.pos0 || fp != st.expected[a.phase] || tc != 0);
        const double bound = sqrt(dd);
        st.fingerprint = fp;
        st.d2_bits = (unsigned long long)__double_as_longlong(dd);
        st.types_changed = tc;
        st.dyn.bound = (bound < 1.0e100) ? bound : -1.0;
        st.dyn.n_shells = tile_shells_for(bound, a.shell_winv);
        st.dyn.stale = stale ? 1u : 0u;
        }
    }

struct AutoPlan
    {
    PairPlan plan;
    // identity of the list
    const uint32_t* nlist = nullptr;
    const uint64_t* head = nullptr;
    const uint32_t* n_neigh = nullptr;
    const double* rcutsq = nullptr;
    uint32_t N = 0, n_max = 0, ntypes = 0;
    bool lanes_one = false;
    int device = -1;
    bool have_plan = false;
    unsigned long long generation = 0;   // caller's list generation at the compile (0: fingerprints)
    double* d_pos0 = nullptr;
    size_t cap_pos0 = 0;
    unsigned long long* d_out = nullptr;
    AutoState* d_state = nullptr;
    AutoState* h_state = nullptr;        // pinned, first AUTO_STATE_HOST_BYTES
    hipStream_t side = nullptr;          // carries the readback: ordered after the check kernel only
    hipEvent_t checked = nullptr;
    float r_list_estimate = 0.f;         // largest listed separation at the last compile
    uint64_t calls = 0;
    uint64_t last_use = 0;
    };

std::mutex g_mutex;
std::vector<std::unique_ptr<AutoPlan>> g_plans;
uint64_t g_clock = 0;
azp_auto_plan_stats g_stats = {0, 0, 0, 0};

void free_auto(AutoPlan& e)
    {
    plan_free(e.plan);
    if (e.d_pos0) (void)hipFree(e.d_pos0);
    if (e.d_out) (void)hipFree(e.d_out);
    if (e.d_state) (void)hipFree(e.d_state);
    if (e.h_state) (void)hipHostFree(e.h_state);
    if (e.side) (void)hipStreamDestroy(e.side);
    if (e.checked) (void)hipEventDestroy(e.checked);
    e.d_pos0 = nullptr; e.d_out = nullptr; e.d_state = nullptr; e.h_state = nullptr; e.side = nullptr; e.checked = nullptr;
    }
} // namespace

bool auto_plan_enabled()
    {
    static const bool on = []
        {
        const char* v = std::getenv("AZP_AUTO_PLAN");
        return !(v && v[0] == '0');
        }();
    return on;
    }

#define AZP_AUTO_TRY(expr)                                   \
    do                                                       \
        {                                                    \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess)                                \
            return (int)e_;                                  \
        } while (0)

namespace
{
void launch_check(const AutoPlan& e, const azp_pair_args& args, bool compare_pos, uint32_t phase, bool learn, hipStream_t stream)
    {
    CheckKArgs k;
    k.pos = args.d_pos;
    k.pos0 = compare_pos ? e.d_pos0 : nullptr;
    k.n_neigh = args.d_n_neigh;
    k.head_list = args.d_head_list;
    k.nlist = args.d_nlist;
    k.rcutsq = args.d_rcutsq;
    k.out = e.d_out;
    k.state = e.d_state;
    k.box = make_box_dev(args.box);
    k.shell_winv = (e.have_plan && e.plan.shell_width > 0.0) ? 1.0 / e.plan.shell_width : 0.0;
    k.generation = args.list_generation;
    k.N = args.N; k.n_max = args.n_max; k.ntypes = args.ntypes;
    // size_nlist = 0 means "unknown": decide from N alone (rows of a pair list hold tens to hundreds of entries)
    k.full = (args.size_nlist ? args.size_nlist <= AUTO_FULL_HASH_ENTRIES : args.N <= 32768u) ? 1u : 0u;
    k.phase = k.full ? 0u : (phase & 7u);
    k.learn = learn ? 1u : 0u;
    k.list_words = args.list_generation ? 0u : 1u;
    if (!k.list_words)
        {
        k.full = 1u;
        k.phase = 0u;
        }
    const uint32_t n_threads = std::max(std::max(args.n_max, args.N), args.ntypes * args.ntypes);
    const uint32_t n_blocks = std::min<uint32_t>(AUTO_CHECK_BLOCKS, (n_threads + 255u) / 256u);
    hipLaunchKernelGGL(auto_check_kernel, dim3(n_blocks), dim3(256), 0, stream, k);
    hipLaunchKernelGGL(auto_fold_kernel, dim3(1), dim3(256), 0, stream, k, n_blocks);
    }

int compile_plan(AutoPlan& e, const azp_pair_args& args, hipStream_t stream)
    {
    azp_pair_args b = args;
    b.range_first = b.range_count = 0;
    b.threads_per_particle = e.lanes_one ? 1u : 0u;
    b.has_displacement_bound = 0;
    b.d_displacement = nullptr;
    // pair_args_t carries no r_cut + r_buff. The shells are cut with the largest listed
    // separation seen at the previous compile (the list radius, to within the last entry
    // inside it); the very first compile for a list learns it and compiles again.
    int passes = 1;
    if (args.r_list_max > 0.0)
        e.plan.shell_hint_r_list = 0.0; // the caller's hint sizes the shells
    else if (e.r_list_estimate > 0.f)
        e.plan.shell_hint_r_list = e.r_list_estimate;
    else
        passes = 2;
    for (int pass = 0; pass < passes; ++pass)
        {
        const int status = plan_build(e.plan, b, stream);
        if (status != AZP_SUCCESS)
            return status;
        ++g_stats.compiles;
        e.r_list_estimate = e.plan.max_listed_r;
        if (passes == 2)
            {
            if (!e.plan.valid || !(e.r_list_estimate > 0.f))
                break;
            e.plan.shell_hint_r_list = e.r_list_estimate;
            }
        }
    if (e.cap_pos0 < (size_t)args.n_max * 4)
        {
        if (e.d_pos0) AZP_AUTO_TRY(hipFree(e.d_pos0));
        e.d_pos0 = nullptr;
        e.cap_pos0 = (size_t)args.n_max * 4 + 1024;
        AZP_AUTO_TRY(hipMalloc(reinterpret_cast<void**>(&e.d_pos0), e.cap_pos0 * sizeof(double)));
        }
    AZP_AUTO_TRY(hipMemcpyAsync(e.d_pos0, args.d_pos, (size_t)args.n_max * 4 * sizeof(double), hipMemcpyDeviceToDevice, stream));
    e.n_max = args.n_max;
    e.generation = args.list_generation;
    e.have_plan = true;
    // learn the fingerprints this list will be compared with (every sample phase of a sampled fingerprint)
    const bool full = args.list_generation || (args.size_nlist ? args.size_nlist <= AUTO_FULL_HASH_ENTRIES : args.N <= 32768u);
    for (uint32_t ph = 0; ph < (full ? 1u : 8u); ++ph)
        launch_check(e, args, false, ph, true, stream);
    AZP_AUTO_TRY(hipGetLastError());
    return AZP_SUCCESS;
    }
} // namespace

int auto_plan_run(const azp_pair_args& args, bool lanes_one, hipStream_t stream, const AutoLauncher& launch_tiled,
                  const std::function<int()>& launch_generic)
    {
    std::lock_guard<std::mutex> lock(g_mutex); // held through the launches: the plan cannot be evicted under a kernel being queued
    int device = 0;
    AZP_AUTO_TRY(hipGetDevice(&device));
    AutoPlan* e = nullptr;
    for (auto& q : g_plans)
        if (q->nlist == args.d_nlist && q->head == args.d_head_list && q->n_neigh == args.d_n_neigh && q->rcutsq == args.d_rcutsq
            && q->N == args.N && q->ntypes == args.ntypes && q->device == device && q->lanes_one == lanes_one)
            e = q.get();
    if (!e)
        {
        if (g_plans.size() >= AUTO_MAX_PLANS)
            {
            size_t oldest = 0;
            for (size_t k = 1; k < g_plans.size(); ++k)
                if (g_plans[k]->last_use < g_plans[oldest]->last_use)
                    oldest = k;
            // the evicted plan's buffers may still be read by a kernel in flight
            AZP_AUTO_TRY(hipDeviceSynchronize());
            free_auto(*g_plans[oldest]);
            g_plans.erase(g_plans.begin() + (long)oldest);
            }
        g_plans.emplace_back(new AutoPlan());
        e = g_plans.back().get();
        e->nlist = args.d_nlist; e->head = args.d_head_list; e->n_neigh = args.d_n_neigh; e->rcutsq = args.d_rcutsq;
        e->N = args.N; e->ntypes = args.ntypes; e->device = device; e->lanes_one = lanes_one;
        AZP_AUTO_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_out), 3 * AUTO_CHECK_BLOCKS * sizeof(unsigned long long)));
        AZP_AUTO_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_state), sizeof(AutoState)));
        AZP_AUTO_TRY(hipMemset(e->d_state, 0, sizeof(AutoState)));
        AZP_AUTO_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_state), sizeof(AutoState), hipHostMallocDefault));
        AZP_AUTO_TRY(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
        AZP_AUTO_TRY(hipEventCreateWithFlags(&e->checked, hipEventDisableTiming));
        }
    e->last_use = ++g_clock;
    ++g_stats.calls;
    const uint64_t call = e->calls++;

    bool stale = true;
    // (a caller that passes list generations: a changed generation needs no check to be known stale; one that
    // stops passing them, or starts to, gets a fresh compile)
    const bool comparable = e->have_plan && e->n_max == args.n_max && e->d_pos0 && args.list_generation == e->generation;
    if (comparable)
        {
        // ---- 1. check kernel; 2. speculative launch right behind it; 3. the host waits for the check alone ----
        launch_check(*e, args, true, (uint32_t)call, false, stream);
        AZP_AUTO_TRY(hipGetLastError());
        AZP_AUTO_TRY(hipEventRecord(e->checked, stream));
        AZP_AUTO_TRY(hipStreamWaitEvent(e->side, e->checked, 0));
        AZP_AUTO_TRY(hipMemcpyAsync(e->h_state, e->d_state, AUTO_STATE_HOST_BYTES, hipMemcpyDeviceToHost, e->side));
        if (e->plan.valid)
            {
            azp_pair_args a = args;
            a.has_displacement_bound = 1;
            a.displacement_bound = 0.0; // (the kernel takes shell count and bound from the device words)
            a.d_displacement = nullptr;
            a.d_stale_flag = nullptr;   // (those belong to callers that run their own distance check)
            a.d_displacement_sq_bits = nullptr;
            const AutoLaunch l = {&e->plan, &a, &e->d_state->dyn};
            const int status = launch_tiled(l);
            if (status != AZP_SUCCESS)
                return status;
            }
        AZP_AUTO_TRY(hipStreamSynchronize(e->side));
        stale = e->h_state->dyn.stale != 0;
        if (!stale)
            {
            ++g_stats.reuses;
            if (e->plan.valid)
                return AZP_SUCCESS; // the speculative launch was the launch
            ++g_stats.generic_fallbacks;
            return launch_generic();
            }
        }
    // ---- 4. (re)compile, launch for good ----
    const int status = compile_plan(*e, args, stream);
    if (status != AZP_SUCCESS)
        return status;
    if (!e->plan.valid)
        {
        ++g_stats.generic_fallbacks;
        return launch_generic();
        }
    azp_pair_args a = args;
    a.has_displacement_bound = 1;
    a.displacement_bound = 0.0; // compiled from these very positions
    a.d_displacement = nullptr;
    a.d_stale_flag = nullptr;
    a.d_displacement_sq_bits = nullptr;
    const AutoLaunch l = {&e->plan, &a, nullptr};
    return launch_tiled(l);
    }

} // namespace azp

extern "C" void azp_pair_auto_plan_clear(void)
    {
    std::lock_guard<std::mutex> lock(azp::g_mutex);
    (void)hipDeviceSynchronize();
    for (auto& q : azp::g_plans)
        azp::free_auto(*q);
    azp::g_plans.clear();
    }

extern "C" void azp_pair_auto_plan_get_stats(azp_auto_plan_stats* out)
    {
    if (!out)
        return;
    std::lock_guard<std::mutex> lock(azp::g_mutex);
    *out = azp::g_stats;
    }
