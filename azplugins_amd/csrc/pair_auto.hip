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

struct CheckKArgs
    {
    const double* pos;
    const double* pos0;       // positions when the plan was compiled (null: no plan yet)
    const uint32_t* n_neigh;
    const uint64_t* head_list;
    const uint32_t* nlist;
    const double* rcutsq;
    unsigned long long* out;  // per workgroup: fingerprint part (sum of mixed words), max |dx|^2 bits, type change flag
    BoxDev box;
    uint32_t N, n_max, ntypes;
    uint32_t full;            // 1: every list entry enters the fingerprint
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

// Grid-stride over the particles; every workgroup leaves ONE partial triple in out[3 * block ..]
// and the host folds the <= AUTO_CHECK_BLOCKS triples after the readback. (Atomics on three
// shared words serialise at ~11 ns each at the memory side: 16k waves took 0.2 ms that way.)
constexpr uint32_t AUTO_CHECK_BLOCKS = 1024;

__global__ void __launch_bounds__(256) auto_check_kernel(const CheckKArgs a)
    {
    unsigned long long h = 0;
    double d2 = 0.0;
    uint32_t type_changed = 0;
    const uint32_t n_items = max(max(a.N, a.n_max), a.ntypes * a.ntypes);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_items; i += gridDim.x * 256u)
        {
        if (i < a.N)
            {
            const uint32_t n = a.n_neigh[i];
            const uint64_t head = a.head_list[i];
            h += mix64(((unsigned long long)i << 32) ^ n) + mix64(head * 0x100000001B3ull + i);
            const uint32_t* row = a.nlist + head;
            if (a.full)
                {
                for (uint32_t k = 0; k < n; ++k)
                    h += mix64(((head + k) << 32) ^ row[k] ^ 0xA5A5A5A500000000ull);
                }
            else if ((i & 7u) == 0 && n)
                h += mix64(((head + n - 1) << 32) ^ row[n - 1]) + mix64(((head + (n >> 1)) << 32) ^ row[n >> 1]);
            }
        if (i < a.ntypes * a.ntypes)
            h += mix64(__double_as_longlong(a.rcutsq[i]) + 0x1234567ull * (i + 1));
        if (i == 0)
            {
            h += mix64(__double_as_longlong(a.box.Lx)) + mix64(__double_as_longlong(a.box.Ly) + 1) + mix64(__double_as_longlong(a.box.Lz) + 2)
                 + mix64(__double_as_longlong(a.box.xy) + 3) + mix64(__double_as_longlong(a.box.xz) + 4)
                 + mix64(__double_as_longlong(a.box.yz) + 5) + mix64(((unsigned long long)a.N << 32) | a.n_max) + mix64(0x77ull + a.ntypes);
            }
        if (a.pos0 && i < a.n_max)
            {
            const double4 p = load_scalar4(a.pos, i), q = load_scalar4(a.pos0, i);
            double dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
            min_image(a.box, dx, dy, dz);
            double d = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
            if (!(d == d))
                d = 1.0e300; // NaN position: whole rows
            d2 = fmax(d2, d);
            type_changed |= (type_from_w(p.w) != type_from_w(q.w)) ? 1u : 0u;
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

struct AutoPlan
    {
    PairPlan plan;
    // identity of the list
    const uint32_t* nlist = nullptr;
    const uint64_t* head = nullptr;
    const uint32_t* n_neigh = nullptr;
    const double* rcutsq = nullptr;
    uint32_t N = 0, n_max = 0, ntypes = 0;
    int device = -1;
    bool have_plan = false;
    unsigned long long fingerprint = 0;
    double* d_pos0 = nullptr;
    size_t cap_pos0 = 0;
    unsigned long long* d_out = nullptr;
    unsigned long long* h_out = nullptr; // pinned
    float r_list_estimate = 0.f;         // largest listed separation at the last compile
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
    if (e.h_out) (void)hipHostFree(e.h_out);
    e.d_pos0 = nullptr; e.d_out = nullptr; e.h_out = nullptr;
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
            {                                                \
            r.status = (int)e_;                              \
            return r;                                        \
            }                                                \
        } while (0)

AutoPlanCheck auto_plan_prepare(const azp_pair_args& args, hipStream_t stream)
    {
    AutoPlanCheck r = {AZP_SUCCESS, nullptr, 0.0};
    std::lock_guard<std::mutex> lock(g_mutex);
    int device = 0;
    AZP_AUTO_TRY(hipGetDevice(&device));
    AutoPlan* e = nullptr;
    for (auto& q : g_plans)
        if (q->nlist == args.d_nlist && q->head == args.d_head_list && q->n_neigh == args.d_n_neigh && q->rcutsq == args.d_rcutsq
            && q->N == args.N && q->ntypes == args.ntypes && q->device == device)
            e = q.get();
    if (!e)
        {
        if (g_plans.size() >= AUTO_MAX_PLANS)
            {
            size_t oldest = 0;
            for (size_t k = 1; k < g_plans.size(); ++k)
                if (g_plans[k]->last_use < g_plans[oldest]->last_use)
                    oldest = k;
            // the evicted plan's buffers may still be read by a kernel in flight on another stream
            AZP_AUTO_TRY(hipDeviceSynchronize());
            free_auto(*g_plans[oldest]);
            g_plans.erase(g_plans.begin() + (long)oldest);
            }
        g_plans.emplace_back(new AutoPlan());
        e = g_plans.back().get();
        e->nlist = args.d_nlist; e->head = args.d_head_list; e->n_neigh = args.d_n_neigh; e->rcutsq = args.d_rcutsq;
        e->N = args.N; e->ntypes = args.ntypes; e->device = device;
        AZP_AUTO_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_out), 3 * AUTO_CHECK_BLOCKS * sizeof(unsigned long long)));
        AZP_AUTO_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_out), 3 * AUTO_CHECK_BLOCKS * sizeof(unsigned long long), hipHostMallocDefault));
        }
    e->last_use = ++g_clock;
    ++g_stats.calls;

    // ---- 1. fingerprint + displacement ----
    const bool compare_pos = e->have_plan && e->n_max == args.n_max && e->d_pos0;
    CheckKArgs k;
    k.pos = args.d_pos;
    k.pos0 = compare_pos ? e->d_pos0 : nullptr;
    k.n_neigh = args.d_n_neigh;
    k.head_list = args.d_head_list;
    k.nlist = args.d_nlist;
    k.rcutsq = args.d_rcutsq;
    k.out = e->d_out;
    k.box = make_box_dev(args.box);
    k.N = args.N; k.n_max = args.n_max; k.ntypes = args.ntypes;
    // size_nlist = 0 means "unknown": decide from N alone (rows of a pair list hold tens to hundreds of entries)
    k.full = (args.size_nlist ? args.size_nlist <= AUTO_FULL_HASH_ENTRIES : args.N <= 32768u) ? 1u : 0u;
    const uint32_t n_threads = std::max(std::max(args.n_max, args.N), args.ntypes * args.ntypes);
    const uint32_t n_blocks = std::min<uint32_t>(AUTO_CHECK_BLOCKS, (n_threads + 255u) / 256u);
    hipLaunchKernelGGL(auto_check_kernel, dim3(n_blocks), dim3(256), 0, stream, k);
    AZP_AUTO_TRY(hipGetLastError());
    // ---- 2. the one readback of the call ----
    AZP_AUTO_TRY(hipMemcpyAsync(e->h_out, e->d_out, 3 * n_blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    AZP_AUTO_TRY(hipStreamSynchronize(stream));
    unsigned long long fp = 0;
    double d2 = 0.0;
    bool types_changed = false;
    for (uint32_t b = 0; b < n_blocks; ++b)
        {
        fp += e->h_out[3 * b];
        double d;
        std::memcpy(&d, &e->h_out[3 * b + 1], sizeof(d));
        d2 = std::max(d2, d);
        types_changed = types_changed || e->h_out[3 * b + 2] != 0;
        }

    // ---- 3. (re)compile ----
    if (!compare_pos || fp != e->fingerprint || types_changed)
        {
        azp_pair_args b = args;
        b.range_first = b.range_count = 0;
        b.threads_per_particle = 0;
        // pair_args_t carries no r_cut + r_buff. The shells are cut with the largest listed
        // separation seen at the previous compile (the list radius, to within the last entry
        // inside it); the very first compile for a list learns it and compiles again.
        int passes = 1;
        if (args.r_list_max > 0.0)
            e->plan.shell_hint_r_list = 0.0; // the caller's hint sizes the shells
        else if (e->r_list_estimate > 0.f)
            e->plan.shell_hint_r_list = e->r_list_estimate;
        else
            passes = 2;
        for (int pass = 0; pass < passes; ++pass)
            {
            r.status = plan_build(e->plan, b, stream);
            if (r.status != AZP_SUCCESS)
                return r;
            ++g_stats.compiles;
            e->r_list_estimate = e->plan.max_listed_r;
            if (passes == 2)
                {
                if (!e->plan.valid || !(e->r_list_estimate > 0.f))
                    break;
                e->plan.shell_hint_r_list = e->r_list_estimate;
                }
            }
        if (e->cap_pos0 < (size_t)args.n_max * 4)
            {
            if (e->d_pos0) AZP_AUTO_TRY(hipFree(e->d_pos0));
            e->d_pos0 = nullptr;
            e->cap_pos0 = (size_t)args.n_max * 4 + 1024;
            AZP_AUTO_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_pos0), e->cap_pos0 * sizeof(double)));
            }
        AZP_AUTO_TRY(hipMemcpyAsync(e->d_pos0, args.d_pos, (size_t)args.n_max * 4 * sizeof(double), hipMemcpyDeviceToDevice, stream));
        e->n_max = args.n_max;
        e->fingerprint = fp;
        e->have_plan = true;
        d2 = 0.0; // the plan was compiled from these very positions
        }
    else
        ++g_stats.reuses;
    if (!e->plan.valid)
        ++g_stats.generic_fallbacks;
    r.plan = &e->plan;
    r.displacement = std::sqrt(d2);
    return r;
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
