// A CPU model of the open-launch protocol (include/robigo_luculenta.h: rl_trace_unit_render; csrc/rl_kernels.hip.h: RlOpenCtl,
// the OPEN variant's refill; csrc/rl_api.hip: session_append / session_begin), for tests/test_open_protocol_model.py.
// TEST INFRASTRUCTURE: it restates the two sides' steps with the same words, the same order and the same memory orders --
//   host:   write entry k; published = k + 1 (seq_cst); read closed_at, final_at:
//           accepted iff closed_at == NONE or closed_at > k; rejected iff final_at != NONE and final_at <= k; else read again.
//   kernel: hand out the jobs below `known`; when there are none left: read published; if it grew, copy the new entries and
//           raise known; otherwise closed_at = known (seq_cst), fence, re-read published: unchanged -> final_at = known and
//           stop; changed -> closed_at = NONE and go on.
// -- and lets real threads race them: several "callers" append jobs as fast as they can (and sometimes pause, so that
// launches run dry), one "kernel" thread per launch consumes them.  Checked: every job is consumed exactly once, by the
// launch that accepted it; no accepted job is left behind when its launch stops; no rejected job is ever consumed.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

namespace {
constexpr uint32_t NONE = 0xffffffffu, CAP = 256;

struct Ctl { // RlOpenCtl
    std::atomic<uint32_t> published{0}, closed_at{NONE}, final_at{NONE};
    std::atomic<uint64_t> jobs[CAP];   // the job's global id (stands for the entry)
    std::atomic<uint32_t> done[CAP];
};

struct Launch {
    Ctl ctl;
    bool open = true; // host view (under the admission lock)
    uint32_t n = 0;
    std::thread kernel;
};

std::vector<std::atomic<uint32_t>> consumed; // per global job id: how often a kernel handed it out
std::atomic<uint64_t> errors{0};

void kernel_main(Launch* l) { // the OPEN kernel's refill path, one "wave"
    Ctl& c = l->ctl;
    uint32_t known = 0, next = 0;
    for (;;) {
        if (next < known) { // hand out job `next`
            const uint64_t id = c.jobs[next].load(std::memory_order_relaxed);
            for (volatile uint32_t spin = 0; spin < (uint32_t)(id * 2654435761u) % 3000u; spin = spin + 1) {} // "tracing": appends arrive meanwhile
            consumed[id].fetch_add(1, std::memory_order_relaxed);
            c.done[next].store(1, std::memory_order_release);
            next += 1;
            continue;
        }
        uint32_t pub = c.published.load(std::memory_order_acquire);
        if (pub > CAP) pub = CAP;
        if (pub > known) { known = pub; continue; }
        // nothing new (the model closes at once: no grace period, which only makes the race harder)
        c.closed_at.store(known, std::memory_order_seq_cst);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        const uint32_t again = c.published.load(std::memory_order_seq_cst);
        if (again == known || known >= CAP) {
            c.final_at.store(known, std::memory_order_seq_cst);
            return;
        }
        c.closed_at.store(NONE, std::memory_order_seq_cst);
    }
}

int append(Launch& l, uint64_t id) { // session_append
    const uint32_t k = l.n;
    l.ctl.jobs[k].store(id, std::memory_order_relaxed);
    l.ctl.done[k].store(0, std::memory_order_relaxed);
    l.ctl.published.store(k + 1, std::memory_order_seq_cst);
    std::atomic_thread_fence(std::memory_order_seq_cst);
    for (;;) {
        const uint32_t closed_at = l.ctl.closed_at.load(std::memory_order_seq_cst);
        if (closed_at == NONE || closed_at > k) break;
        const uint32_t final_at = l.ctl.final_at.load(std::memory_order_seq_cst);
        if (final_at != NONE && final_at <= k) return -1;
    }
    l.n = k + 1;
    return (int)k;
}
} // namespace

int main(int argc, char** argv) {
    const unsigned callers = argc > 1 ? (unsigned)atoi(argv[1]) : 4, per_caller = argc > 2 ? (unsigned)atoi(argv[2]) : 20000;
    consumed = std::vector<std::atomic<uint32_t>>(callers * per_caller);
    std::mutex admission;
    std::vector<Launch*> launches;
    Launch* current = nullptr;
    uint64_t rejected_total = 0;
    auto caller = [&](unsigned me) {
        std::mt19937 rng(me + 1);
        for (unsigned i = 0; i < per_caller; ++i) {
            const uint64_t id = (uint64_t)me * per_caller + i;
            Launch* mine = nullptr;
            int k = -1;
            {
                std::lock_guard<std::mutex> guard(admission);
                for (;;) {
                    if (current && current->open && current->n < CAP) {
                        k = append(*current, id);
                        if (k >= 0) { mine = current; break; }
                        current->open = false;
                        current->kernel.join(); // it stopped before it saw the entry: final_at is set, the thread is on its way out
                        rejected_total += 1;
                        continue;
                    }
                    if (current && current->kernel.joinable()) { // a launch whose job table is full: no append will close it
                        current->open = false;                    // (the model waits for it; the library uses another slot meanwhile)
                        current->kernel.join();
                    }
                    current = new Launch(); // session_start: the call is job 0 of a new launch
                    launches.push_back(current);
                    current->ctl.jobs[0].store(id, std::memory_order_relaxed);
                    current->ctl.done[0].store(0, std::memory_order_relaxed);
                    current->ctl.published.store(1, std::memory_order_seq_cst);
                    current->n = 1;
                    current->kernel = std::thread(kernel_main, current);
                    mine = current;
                    k = 0;
                    break;
                }
            }
            const auto t0 = std::chrono::steady_clock::now(); // session_wait
            while (mine->ctl.done[k].load(std::memory_order_acquire) == 0) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
                    errors += 1; // accepted, never consumed
                    std::fprintf(stderr, "job %llu: accepted as %d of a launch that never handed it out\n", (unsigned long long)id, k);
                    break;
                }
                std::this_thread::yield();
            }
            if ((rng() & 127u) == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 200)); // let launches run dry
        }
    };
    std::vector<std::thread> threads;
    for (unsigned c = 0; c < callers; ++c) threads.emplace_back(caller, c);
    for (auto& t : threads) t.join();
    for (Launch* l : launches)
        if (l->kernel.joinable()) l->kernel.join();
    uint64_t accepted = 0;
    for (Launch* l : launches) {
        const uint32_t final_at = l->ctl.final_at.load();
        if (final_at != l->n) errors += 1, std::fprintf(stderr, "a launch stopped at %u with %u accepted jobs\n", final_at, l->n);
        accepted += l->n;
        delete l;
    }
    for (size_t id = 0; id < consumed.size(); ++id)
        if (consumed[id].load() != 1) errors += 1, std::fprintf(stderr, "job %zu handed out %u times\n", id, consumed[id].load());
    if (accepted != consumed.size()) errors += 1;
    std::printf("%s: %zu jobs from %u callers over %zu launches, %llu appends turned down by a closing launch, %llu errors\n",
                errors.load() == 0 ? "ok" : "FAILED", consumed.size(), callers, launches.size(), (unsigned long long)rejected_total,
                (unsigned long long)errors.load());
    return errors.load() == 0 ? 0 : 1;
}
