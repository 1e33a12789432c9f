// rl_app.cpp -- App (app.rs:48-164): the worker pool that asks the TaskScheduler for tasks and
// executes them, with the GPU units of the C ABI where the reference has CPU loops.
//
// Same structure as the reference: `concurrency` OS threads, one Mutex around the scheduler
// (app.rs:57,107), tasks executed outside the lock.  Differences, forced by "the reference never
// terminates and is unseedable": a batch budget (max_batches), a path range per trace task, explicit
// checkpoint / image file names; and forced by the time scale (a task takes a millisecond, not
// seconds): Task::Sleep waits 200 us instead of 100 ms (app.rs:129; RlAppConfig::sleep_us), and in
// fused mode a Plot task renders the batches of its trace units as one launch.
//
// Several GPUs in one process (RlAppConfig::n_devices > 1): every unit of the scheduler is a LOGICAL unit
// with one physical unit per rank (= per GPU, or per RNG stream when a device is listed twice).  A Trace or
// Plot task runs on all ranks at once -- same batch indices, RNG stream = stream + rank, so the ranks' samples
// are disjoint -- and a Gather task first sums the ranks' plot buffers onto rank 0 (ranks sharing a device by
// a device-local add, distinct devices by one grouped ncclReduce over xGMI: rl_plot_unit_reduce) and then
// Kahan-accumulates on rank 0 (gather_unit.rs:49-64).  One scheduler, the reference's protocol unchanged, and
// the collective is in lockstep by construction because one worker thread issues it for all ranks.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <cerrno>
#include <fcntl.h>  // open(O_DIRECTORY): the fsync of a directory behind a rename
#include <unistd.h> // fsync

#include "../../include/robigo_luculenta.h"

void rl_internal_set_last_error(const std::string& msg); // rl_api.hip

namespace {

struct Rank { // one GPU (or one RNG stream of a GPU that is listed twice)
    int device = 0;
    int leader = 0;            // first rank on the same device
    RlScene* scene = nullptr;
    RlComm* comm = nullptr;    // set on device leaders when there is more than one distinct device
    std::vector<RlTraceUnit*> trace_units;
    std::vector<RlPlotUnit*> plot_units;
};

// One line of the index: the buffer holds every path below next_batch * photons_per_batch of `ranks` RNG streams from `stream` on,
// seed `seed`.  A buffer that was continued with another seed, or with streams it had not seen, holds several such sets;
// the index keeps them ALL (rewriting it with only the current run's set forgot the older samples, and a later
// resume with the original seed would have added them again).
struct ResumeEntry {
    unsigned long long next = 0, seed = 0;
    unsigned ppb = 0, stream = 0;
    size_t ranks = 0;
};

struct AppState {
    const RlAppConfig* cfg;
    RlScheduler* scheduler = nullptr;
    std::vector<Rank> ranks;
    bool use_rccl = false;     // more than one distinct device
    bool distinct_devices = true; // no device is listed twice
    RlGatherUnit* gather = nullptr;   // on rank 0's device
    RlTonemapUnit* tonemap = nullptr; // on rank 0's device
    uint64_t first_batch = 0;
    std::vector<ResumeEntry> other_samples; // resume: what the loaded buffer holds of other seeds / RNG streams (kept in the index)
    std::mutex lock;                       // Arc<Mutex<TaskScheduler>>, app.rs:57
    std::mutex lock_counters;              // traces_issued, read by save_checkpoint outside `lock`
    std::chrono::steady_clock::time_point t0;
    uint64_t traces_issued = 0;            // under `lock`
    std::atomic<uint64_t> fused_next_path{0}; // fused mode: next unrendered path index
    std::vector<uint64_t> trace_first_path; // per trace unit: the path range of its current task
    uint64_t tasks[5] = {0, 0, 0, 0, 0};
    uint32_t tonemaps = 0;
    std::atomic<int> error{0};
    std::string error_message;
    std::vector<uint8_t> rgb;
    uint32_t photons;
};

int64_t now_ms(const AppState& a) {
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - a.t0).count();
}

void fail(AppState& a, int rc) {
    int expected = 0;
    if (a.error.compare_exchange_strong(expected, rc)) a.error_message = rl_last_error();
}

// Minimal PNG encoder (RGB8, zlib "stored" blocks, no external library) so the App can write the
// reference's output.png (main.rs:61) -- the image crate only matters for the file format.
uint32_t crc32_update(uint32_t crc, const uint8_t* data, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ data[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    std::vector<uint8_t> body(type, type + 4);
    body.insert(body.end(), data.begin(), data.end());
    out.insert(out.end(), body.begin(), body.end());
    put_be32(out, crc32_update(0xffffffffu, body.data(), body.size()) ^ 0xffffffffu);
}

int write_png(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw; // filter byte 0 + row
    raw.reserve((size_t)h * (w * 3 + 1));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0; // adler32
    for (size_t i = 0; i < raw.size(); ++i) {
        a = (a + raw[i]) % 65521u;
        b = (b + a) % 65521u;
    }
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (n == 0) break;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    const uint8_t tail[5] = {8, 2, 0, 0, 0}; // 8-bit RGB, deflate, no filter, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", std::vector<uint8_t>());
    FILE* f = std::fopen(path, "wb");
    if (!f) return RL_E_IO;
    const size_t written = std::fwrite(out.data(), 1, out.size(), f);
    return (std::fclose(f) == 0 && written == out.size()) ? RL_OK : RL_E_IO;
}

int write_image(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h);

int write_ppm(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return RL_E_IO;
    std::fprintf(f, "P6\n%u %u\n255\n", w, h);
    const size_t n = (size_t)w * h * 3;
    const size_t written = std::fwrite(rgb, 1, n, f);
    return (std::fclose(f) == 0 && written == n) ? RL_OK : RL_E_IO;
}

int write_image(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h) {
    const size_t n = std::strlen(path);
    if (n >= 4 && std::strcmp(path + n - 4, ".png") == 0) return write_png(path, rgb, w, h);
    return write_ppm(path, rgb, w, h);
}

// buffer.raw (gather_unit.rs:68-79) plus a sidecar "<checkpoint>.next" holding the index of the first batch a
// resumed run must render.  The reference needs no such thing -- every restart draws fresh numbers from its
// OS-seeded generator -- but here a path is a pure function of (seed, stream, path index), so a resumed run that
// started at batch 0 again would add the very same samples twice: brighter, not less noisy.  The index written
// is one past the highest batch handed out so far; batches still in flight when the file is written are
// skipped on resume, never repeated.
std::string sidecar_path(const char* checkpoint) { return std::string(checkpoint) + ".next"; }

// Both files are replaced atomically (written beside themselves, then renamed), the index FIRST: a crash between the
// two leaves a newer index beside an older buffer -- the samples in between are lost, which is noise, whereas the other
// order would add them twice on resume, which is bias.
// The rename itself is made durable too: a rename lives in the directory, and without an fsync of the directory the
// kernel may persist two renames in either order -- the buffer's could reach the disk with the old index still in place, and a
// resume after a power loss would then add those samples a second time.
int replace_file(const std::string& tmp, const std::string& path) {
    if (std::rename(tmp.c_str(), path.c_str()) != 0) return RL_E_IO;
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? std::string(".") : (slash == 0 ? std::string("/") : path.substr(0, slash));
    // (the rename HAS happened; the directory's fsync is best effort where a file system does not offer it -- some NFS,
    // CIFS and FUSE mounts answer EINVAL / ENOTSUP, a read-only bind EROFS -- and only a real I/O error fails the save)
    auto soft = [](int e) { return e == EINVAL || e == ENOTSUP || e == EROFS || e == EACCES || e == EBADF; };
    int fd;
    do fd = open(dir.c_str(), O_RDONLY | O_DIRECTORY); while (fd < 0 && errno == EINTR);
    if (fd < 0) return soft(errno) ? RL_OK : RL_E_IO;
    int rc;
    do rc = fsync(fd); while (rc != 0 && errno == EINTR);
    const int e = errno;
    close(fd);
    return (rc == 0 || soft(e)) ? RL_OK : RL_E_IO;
}

std::vector<ResumeEntry> read_sidecar(const char* checkpoint, uint32_t photons, const RlAppConfig* config, size_t ranks) {
    std::vector<ResumeEntry> entries;
    FILE* g = std::fopen(sidecar_path(checkpoint).c_str(), "r");
    if (!g) return entries; // a buffer without an index (a reference run's buffer.raw): nothing to continue from
    for (;;) {
        ResumeEntry e;
        const int got = std::fscanf(g, " next_batch %llu photons_per_batch %u seed %llu stream %u ranks %zu", &e.next, &e.ppb, &e.seed, &e.stream, &e.ranks);
        if (got < 1) break;
        if (got < 5) { // an index written by an older build: batches of the current size, the current streams
            e.ppb = photons, e.seed = config->seed, e.stream = config->stream, e.ranks = ranks;
            entries.push_back(e);
            break;
        }
        entries.push_back(e);
    }
    std::fclose(g);
    return entries;
}

int save_checkpoint(AppState& a) {
    uint64_t next;
    {
        std::lock_guard<std::mutex> guard(a.lock_counters);
        next = a.first_batch + a.traces_issued;
    }
    const std::string side = sidecar_path(a.cfg->checkpoint), side_tmp = side + ".tmp";
    FILE* f = std::fopen(side_tmp.c_str(), "w");
    if (!f) return RL_E_IO;
    std::fprintf(f, "next_batch %llu\nphotons_per_batch %u\nseed %llu\nstream %u\nranks %zu\n", (unsigned long long)next, a.photons,
                 (unsigned long long)a.cfg->seed, a.cfg->stream, a.ranks.size());
    for (const ResumeEntry& e : a.other_samples) // what the buffer already held of other seeds / streams when this run loaded it
        std::fprintf(f, "next_batch %llu\nphotons_per_batch %u\nseed %llu\nstream %u\nranks %zu\n", e.next, e.ppb, e.seed, e.stream, e.ranks);
    // on disk before the rename makes it the index, and the rename on disk (replace_file) before the buffer is touched: the
    // index-before-buffer order must survive a power loss, not only a crash
    const bool flushed = std::fflush(f) == 0 && fsync(fileno(f)) == 0;
    if (std::fclose(f) != 0 || !flushed) return RL_E_IO;
    int rc = replace_file(side_tmp, side);
    if (rc != RL_OK) return rc;
    const std::string buf = a.cfg->checkpoint, buf_tmp = buf + ".tmp";
    rc = rl_gather_unit_save(a.gather, buf_tmp.c_str()); // (flushes and fsyncs before it closes the file)
    return rc != RL_OK ? rc : replace_file(buf_tmp, buf);
}

// Where a resumed run starts.  The index records what each of its sets counts (paths per batch) and which samples it stands for
// (seed, first RNG stream, number of ranks = streams).  A set with this run's seed and streams: continue behind its path index,
// whatever the batch size is now.  Sets of another seed, or of streams this run does not use: every sample of this run is new, the
// configured first_batch stands -- and the sets are kept (`others`) so that the next index still lists them.  A set of this seed
// whose streams only partly overlap this run's: some ranks would repeat samples and others not -- refused.
int resume_first_batch(const RlAppConfig* config, uint32_t photons, size_t ranks, uint64_t* first_batch, std::vector<ResumeEntry>* others, std::string* err) {
    for (const ResumeEntry& e : read_sidecar(config->checkpoint, photons, config, ranks)) {
        const uint64_t lo = config->stream, hi = lo + ranks, olo = e.stream, ohi = olo + e.ranks;
        if (e.seed != config->seed || hi <= olo || ohi <= lo) { // another seed, or disjoint streams
            others->push_back(e);
            continue;
        }
        if (lo != olo || hi != ohi) {
            *err = "resume: the checkpoint holds RNG streams [" + std::to_string(olo) + ", " + std::to_string(ohi) + ") of this seed, the run asks for [" +
                   std::to_string(lo) + ", " + std::to_string(hi) + "): some ranks would repeat samples";
            return RL_E_STATE;
        }
        const unsigned __int128 paths = (unsigned __int128)e.next * e.ppb;
        const uint64_t batch = (uint64_t)((paths + photons - 1) / photons); // the first batch of the current size that starts behind them
        if (batch > *first_batch) *first_batch = batch;
    }
    return RL_OK;
}

// App::execute_task (app.rs:113-126)
void execute_task(AppState& a, const RlTask& task) {
    const RlAppConfig& c = *a.cfg;
    int rc = RL_OK;
    switch (task.kind) {
    case RL_TASK_SLEEP: // app.rs:128-130, at the time scale of GPU tasks (RlAppConfig::sleep_us)
        std::this_thread::sleep_for(std::chrono::microseconds(c.sleep_us ? c.sleep_us : 200u));
        break;
    case RL_TASK_TRACE: // app.rs:132-134, on every rank at once
        if (!c.fused) {
            // A Trace task BEGINS TraceUnit::render (rl_trace_unit_render_begin: the call joins the device's open launch,
            // rl_api.hip) and the worker moves on; the Plot task that takes the unit ends it (rl_plot_unit_plot waits for the
            // photons it plots).  With 3 x concurrency trace units in circulation that keeps the device fed even by one or
            // two workers (measured un-fused at 1 / 4 / 8 workers: 7.2 / 12.2 / 13.0 Grays/s, against 3.7 / 8.6 / 11.1 when
            // the Trace task waits for its own batch like a reference worker: RlAppConfig::blocking_trace).
            // Ranks that share a device (a test layout) would only take turns at it with their open launches (one kernel
            // serves one RNG stream): they get a plain launch per batch each instead.
            for (size_t r = 0; r < a.ranks.size() && rc == RL_OK; ++r) {
                RlTraceUnit* u = a.ranks[r].trace_units[task.unit];
                if (a.distinct_devices) {
                    rc = rl_trace_unit_render_begin(u, a.ranks[r].scene, c.seed, c.stream + (uint32_t)r, a.trace_first_path[task.unit]);
                } else {
                    rc = rl_trace_unit_sync(u); // back-pressure: this unit's previous launch (long finished, normally)
                    if (rc == RL_OK)
                        rc = rl_trace_unit_render_async(u, a.ranks[r].scene, c.seed, c.stream + (uint32_t)r, a.trace_first_path[task.unit]);
                }
            }
            if (c.blocking_trace)
                for (size_t r = 0; r < a.ranks.size(); ++r) { // every begun call is ended, also after an error
                    const int rc_end = rl_trace_unit_sync(a.ranks[r].trace_units[task.unit]);
                    if (rc == RL_OK) rc = rc_end;
                }
        }
        // fused: the photons are produced when the unit is plotted (the target buffer is known then)
        break;
    case RL_TASK_PLOT: // app.rs:136-141
        if (!c.fused) {
            for (size_t r = 0; r < a.ranks.size() && rc == RL_OK; ++r) {
                std::vector<RlTraceUnit*> units;
                for (uint32_t i = 0; i < task.n_units; ++i) units.push_back(a.ranks[r].trace_units[task.units[i]]);
                rc = rl_plot_unit_plot(a.ranks[r].plot_units[task.unit], units.data(), (uint32_t)units.size());
            }
        } else {
            // The batches of all the trace units of this task go out as ONE launch over one contiguous
            // range of path indices: a 524,288-path launch of its own would keep an MI355X busy for 0.15 ms
            // and spend twice that waiting for its longest paths, n of them together amortise that tail.
            // Ranges are handed out in execution order, so every index below traces_issued * photons is
            // rendered exactly once whichever worker plots which units.
            if (task.n_units != 0) {
                const uint64_t n = (uint64_t)task.n_units * (uint64_t)a.photons;
                const uint64_t first = a.fused_next_path.fetch_add(n);
                for (size_t r = 0; r < a.ranks.size() && rc == RL_OK; ++r) {
                    RlTraceUnit* u = a.ranks[r].trace_units[task.units[0]];
                    if (a.distinct_devices) { // begun here, ended by the Gather task that takes the plot unit (as above)
                        rc = rl_trace_unit_render_fused_begin(u, a.ranks[r].scene, a.ranks[r].plot_units[task.unit], c.seed,
                                                              c.stream + (uint32_t)r, first, n);
                    } else {
                        rc = rl_trace_unit_render_fused(u, a.ranks[r].scene, a.ranks[r].plot_units[task.unit], c.seed,
                                                        c.stream + (uint32_t)r, first, n);
                    }
                }
                if (c.blocking_trace)
                    for (size_t r = 0; r < a.ranks.size(); ++r) {
                        const int rc_end = rl_plot_unit_sync(a.ranks[r].plot_units[task.unit]);
                        if (rc == RL_OK) rc = rc_end;
                    }
            }
        }
        break;
    case RL_TASK_GATHER: // app.rs:143-152 (save moved to tonemap time)
        for (uint32_t i = 0; i < task.n_units && rc == RL_OK; ++i) {
            const uint32_t j = task.units[i];
            // ranks that share a GPU: a device-local sum onto the device's first rank
            for (size_t r = 0; r < a.ranks.size() && rc == RL_OK; ++r)
                if (a.ranks[r].leader != (int)r) {
                    rc = rl_plot_unit_add(a.ranks[a.ranks[r].leader].plot_units[j], a.ranks[r].plot_units[j]);
                    if (rc == RL_OK) rc = rl_plot_unit_clear(a.ranks[r].plot_units[j]);
                }
            // distinct GPUs: one grouped ncclReduce onto rank 0 over xGMI
            if (a.use_rccl && rc == RL_OK) {
                rc = rl_comm_group_start();
                for (size_t r = 0; r < a.ranks.size() && rc == RL_OK; ++r)
                    if (a.ranks[r].comm) rc = rl_plot_unit_reduce(a.ranks[r].plot_units[j], a.ranks[r].comm, 0);
                const int end_rc = rl_comm_group_end();
                if (rc == RL_OK) rc = end_rc;
                for (size_t r = 1; r < a.ranks.size() && rc == RL_OK; ++r)
                    if (a.ranks[r].comm) rc = rl_plot_unit_clear(a.ranks[r].plot_units[j]);
            }
            if (rc == RL_OK) rc = rl_gather_unit_accumulate(a.gather, a.ranks[0].plot_units[j]); // Kahan + clear
        }
        break;
    case RL_TASK_TONEMAP: // app.rs:154-164
        rc = rl_tonemap_unit_tonemap(a.tonemap, a.gather);
        if (rc == RL_OK) rc = rl_tonemap_unit_rgb(a.tonemap, a.rgb.data());
        if (rc == RL_OK && c.output_ppm) rc = write_image(c.output_ppm, a.rgb.data(), c.width, c.height);
        if (rc == RL_OK && c.checkpoint) rc = save_checkpoint(a);
        if (rc == RL_OK && c.verbose && c.output_ppm) std::printf("wrote image to %s\n", c.output_ppm);
        break;
    default: break;
    }
    if (rc != RL_OK) fail(a, rc);
}

void log_completed(const AppState& a, const RlTask& t) { // task_scheduler.rs:241-296
    switch (t.kind) {
    case RL_TASK_TRACE: std::printf("done tracing with unit %u\n", t.unit); break;
    case RL_TASK_PLOT:
        std::printf("done plotting with unit %u\nthe following trace units are available again: ", t.unit);
        for (uint32_t i = 0; i < t.n_units; ++i) std::printf(" %u ", t.units[i]);
        std::printf("\n");
        break;
    case RL_TASK_GATHER:
        std::printf("done gathering\nthe following plot units are available again: ");
        for (uint32_t i = 0; i < t.n_units; ++i) std::printf(" %u ", t.units[i]);
        std::printf("\n");
        break;
    case RL_TASK_TONEMAP: std::printf("done tonemapping\n"); break;
    default: break;
    }
    (void)a;
}

// Who is who when one process drives several ranks, as a function a test can call without a GPU (rl_debug_app_rank_plan):
// rank r renders on rank_device[r] with RNG stream `stream + r`; leader[r] is the first rank on the same device (ranks that share a GPU
// are summed onto it with rl_plot_unit_add before anything crosses xGMI); comm_rank[r] is the rank's place in the RCCL communicator --
// one per DISTINCT device, in order of first appearance, so rank 0's device is communicator rank 0, the root of every reduce -- or -1
// for a rank that has none (not its device's leader, or a run on one device).  Returns the distinct devices in communicator order.
std::vector<int> plan_ranks(int device, const int* devices, uint32_t n_devices, int* rank_device, int* leader, int* comm_rank) {
    const uint32_t n_ranks = n_devices > 1 ? n_devices : 1;
    std::vector<int> distinct;
    for (uint32_t r = 0; r < n_ranks; ++r) {
        rank_device[r] = n_devices > 1 ? (devices ? devices[r] : device + (int)r) : device;
        leader[r] = (int)r;
        for (uint32_t q = 0; q < r; ++q)
            if (rank_device[q] == rank_device[r]) {
                leader[r] = leader[q];
                break;
            }
        if (leader[r] == (int)r) distinct.push_back(rank_device[r]);
    }
    size_t next = 0;
    for (uint32_t r = 0; r < n_ranks; ++r) comm_rank[r] = (distinct.size() > 1 && leader[r] == (int)r) ? (int)next++ : -1;
    return distinct;
}

// App::start_worker's loop (app.rs:92-111), ending when the batch budget is exhausted.
void worker(AppState* ap) {
    AppState& a = *ap;
    RlTask task;
    std::memset(&task, 0, sizeof task);
    task.kind = RL_TASK_SLEEP; // "this worker is done sleeping" (app.rs:98-99)
    for (;;) {
        RlTask next;
        {
            std::lock_guard<std::mutex> guard(a.lock);
            if (a.cfg->verbose) log_completed(a, task);
            const bool budget_left = a.traces_issued < a.cfg->max_batches;
            if (rl_scheduler_get_new_task(a.scheduler, &task, now_ms(a), &next) != RL_OK) {
                fail(a, RL_E_INVALID);
                return;
            }
            if (next.kind == RL_TASK_TRACE) {
                if (!budget_left) {
                    // Out of budget: hand the unit straight back as an un-rendered no-op is not possible in
                    // the reference's protocol, so leave it out of circulation and stop this worker.
                    return;
                }
                a.trace_first_path[next.unit] = (a.first_batch + a.traces_issued) * (uint64_t)a.photons;
                std::lock_guard<std::mutex> counters(a.lock_counters);
                a.traces_issued += 1;
            }
            if (next.kind == RL_TASK_TONEMAP) a.tonemaps += 1;
            a.tasks[next.kind] += 1;
            if (a.cfg->verbose && next.kind == RL_TASK_TONEMAP) {
                float mean = 0, sd = 0;
                if (rl_scheduler_performance(a.scheduler, &mean, &sd) == RL_OK && mean == mean)
                    std::printf("performance: %g +- %g batches/sec\n", mean, sd);
            }
        }
        execute_task(a, next);
        if (a.error.load() != 0) return;
        task = next;
    }
}

} // namespace

extern "C" int rl_app_run(const RlAppConfig* config, RlAppStats* stats, uint8_t* rgb_out) {
    if (!config || config->width == 0 || config->height == 0 || config->concurrency == 0) {
        rl_internal_set_last_error("rl_app_run: null config, zero-sized image or zero workers");
        return RL_E_INVALID;
    }
    if (config->concurrency * 3 > RL_TASK_MAX_UNITS) {
        rl_internal_set_last_error("rl_app_run: concurrency " + std::to_string(config->concurrency) + " needs " +
                                   std::to_string(config->concurrency * 3) + " trace units, more than RL_TASK_MAX_UNITS = " +
                                   std::to_string(RL_TASK_MAX_UNITS) + " (at most " + std::to_string(RL_TASK_MAX_UNITS / 3) + " workers)");
        return RL_E_INVALID;
    }
    AppState a;
    a.cfg = config;
    a.t0 = std::chrono::steady_clock::now(); // also the origin of `seconds` when set-up fails
    a.photons = config->photons_per_batch ? config->photons_per_batch : 1024u * 512u;
    a.rgb.assign((size_t)config->width * config->height * 3, 0);
    const uint32_t n_trace = config->concurrency * 3;                          // task_scheduler.rs:95
    const uint32_t n_plot = config->concurrency / 2 > 1 ? config->concurrency / 2 : 1; // :96
    int rc = rl_scheduler_create(config->concurrency, config->tonemap_interval_ms, &a.scheduler);
    if (rc != RL_OK) rl_internal_set_last_error("rl_app_run: rl_scheduler_create failed");

    // The ranks: one per listed device (a device may be listed more than once: several RNG streams on one GPU).
    const uint32_t n_ranks = config->n_devices > 1 ? config->n_devices : 1;
    a.ranks.resize(n_ranks);
    std::vector<int> rank_device(n_ranks), leader(n_ranks), comm_rank(n_ranks);
    const std::vector<int> distinct = plan_ranks(config->device, config->devices, config->n_devices, rank_device.data(), leader.data(), comm_rank.data());
    for (uint32_t r = 0; r < n_ranks; ++r) a.ranks[r].device = rank_device[r], a.ranks[r].leader = leader[r];
    a.use_rccl = distinct.size() > 1;
    a.distinct_devices = distinct.size() == n_ranks;
    if (rc == RL_OK && a.use_rccl) {
        std::vector<RlComm*> comms(distinct.size(), nullptr);
        rc = rl_comm_init_all(distinct.data(), (int)distinct.size(), comms.data()); // rank 0's device is distinct[0] = comm rank 0
        for (uint32_t r = 0; r < n_ranks; ++r)
            if (comm_rank[r] >= 0) a.ranks[r].comm = comms[(size_t)comm_rank[r]];
    }

    std::vector<RlObjectDesc> objects;
    RlCameraDesc camera;
    uint32_t n_objects = 0;
    if (rc == RL_OK) {
        rl_scene_builtin_desc(config->builtin_scene, config->builtin_param, nullptr, 0, &n_objects, &camera);
        objects.resize(n_objects);
        rc = rl_scene_builtin_desc(config->builtin_scene, config->builtin_param, objects.data(), n_objects, &n_objects, &camera);
    }
    for (uint32_t r = 0; r < n_ranks && rc == RL_OK; ++r) {
        Rank& k = a.ranks[r];
        RlSceneDesc desc;
        desc.n_objects = n_objects;
        desc.objects = objects.data();
        desc.camera = camera;
        rc = rl_scene_create(&desc, k.device, &k.scene); // Arc::new(App::set_up_scene()), app.rs:63: one copy per rank
        for (uint32_t i = 0; i < n_trace && rc == RL_OK; ++i) {
            RlTraceUnit* u = nullptr;
            rc = rl_trace_unit_create(k.device, i, config->width, config->height, a.photons, &u);
            if (u) k.trace_units.push_back(u);
        }
        for (uint32_t i = 0; i < n_plot && rc == RL_OK; ++i) {
            RlPlotUnit* u = nullptr;
            rc = rl_plot_unit_create(k.device, i, config->width, config->height, nullptr, &u);
            if (u) k.plot_units.push_back(u);
        }
    }
    const int root_device = a.ranks[0].device;
    if (rc == RL_OK) rc = rl_gather_unit_create(root_device, config->width, config->height, &a.gather);
    if (rc == RL_OK) rc = rl_tonemap_unit_create(root_device, config->width, config->height, &a.tonemap);
    a.first_batch = config->first_batch;
    if (rc == RL_OK && config->resume && config->checkpoint) {
        FILE* f = std::fopen(config->checkpoint, "rb"); // a missing file is not an error (gather_unit.rs:82)
        if (f) {
            std::fclose(f);
            rc = rl_gather_unit_load(a.gather, config->checkpoint);
            // Continue where the checkpointed run stopped handing out batches (see save_checkpoint).
            std::string why;
            if (rc == RL_OK && (rc = resume_first_batch(config, a.photons, a.ranks.size(), &a.first_batch, &a.other_samples, &why)) != RL_OK) rl_internal_set_last_error(why);
        }
    }
    a.fused_next_path = a.first_batch * (uint64_t)a.photons;
    a.trace_first_path.assign(n_trace, 0);

    if (rc == RL_OK) {
        a.t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        // app.rs:66-70 starts one thread per unit of scheduler depth; here the two are separate knobs (RlAppConfig::threads)
        const uint32_t n_threads = config->threads ? (config->threads < config->concurrency ? config->threads : config->concurrency) : config->concurrency;
        for (uint32_t i = 0; i < n_threads; ++i) pool.emplace_back(worker, &a);
        for (std::thread& t : pool) t.join();
        rc = a.error.load();
    }
    // Drain: every traced unit is plotted, every plot gathered, one final tonemap -- the same tasks the
    // scheduler would have issued, run on this thread (workers stopped when the budget ran out).
    if (rc == RL_OK) {
        RlTask task;
        std::memset(&task, 0, sizeof task);
        task.kind = RL_TASK_SLEEP;
        for (int guard = 0; guard < 100000; ++guard) {
            RlTask next;
            if (rl_scheduler_get_new_task(a.scheduler, &task, now_ms(a), &next) != RL_OK) {
                rc = RL_E_INVALID;
                break;
            }
            if (next.kind == RL_TASK_SLEEP) break; // nothing left to plot or gather
            if (next.kind == RL_TASK_TRACE) {
                // The scheduler prefers new traces over plotting (task_scheduler.rs:160-169); park the
                // idle unit (never handed back) and ask again until only plot / gather work is left.
                task.kind = RL_TASK_SLEEP;
                task.n_units = 0;
                continue;
            }
            a.tasks[next.kind] += 1;
            if (next.kind == RL_TASK_TONEMAP) a.tonemaps += 1;
            execute_task(a, next);
            if ((rc = a.error.load()) != RL_OK) break;
            task = next;
        }
    }
    if (rc == RL_OK) {
        RlTask final_task;
        std::memset(&final_task, 0, sizeof final_task);
        final_task.kind = RL_TASK_TONEMAP;
        a.tasks[RL_TASK_TONEMAP] += 1;
        a.tonemaps += 1;
        execute_task(a, final_task);
        rc = a.error.load();
    }
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - a.t0).count();
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->batches = a.traces_issued;
        stats->next_batch = a.first_batch + a.traces_issued;
        stats->seconds = seconds;
        stats->tonemaps = a.tonemaps;
        for (int k = 0; k < 5; ++k) stats->tasks[k] = a.tasks[k];
        for (Rank& k : a.ranks)
            for (RlTraceUnit* u : k.trace_units) {
                uint64_t p = 0, s = 0;
                double ms = 0;
                if (rl_trace_unit_stats(u, &p, &s, &ms) == RL_OK) {
                    stats->paths += p;
                    stats->segments += s;
                    stats->kernel_ms += ms;
                }
            }
        if (a.scheduler) rl_scheduler_performance(a.scheduler, &stats->batches_per_sec_mean, &stats->batches_per_sec_stddev);
    }
    if (rgb_out && rc == RL_OK) std::memcpy(rgb_out, a.rgb.data(), a.rgb.size());
    std::string message = a.error_message;
    if (rc != RL_OK && message.empty()) message = rl_last_error(); // a set-up call on this thread failed
    for (Rank& k : a.ranks) {
        for (RlTraceUnit* u : k.trace_units) rl_trace_unit_destroy(u);
        for (RlPlotUnit* u : k.plot_units) rl_plot_unit_destroy(u);
        rl_scene_destroy(k.scene);
    }
    rl_gather_unit_destroy(a.gather);
    rl_tonemap_unit_destroy(a.tonemap);
    for (Rank& k : a.ranks) rl_comm_destroy(k.comm);
    rl_scheduler_destroy(a.scheduler);
    if (rc != RL_OK && !message.empty()) rl_internal_set_last_error(message); // the failing call may have run on a worker thread
    return rc;
}

// Diagnostics (robigo_luculenta_debug.h): the rank bookkeeping of rl_app_run for a device list, no GPU needed.
extern "C" int rl_debug_app_rank_plan(int device, const int* devices, uint32_t n_devices, int* rank_device, int* leader, int* comm_rank,
                                      uint32_t* n_communicator_ranks) {
    if (!rank_device || !leader || !comm_rank || !n_communicator_ranks) {
        rl_internal_set_last_error("rl_debug_app_rank_plan: null output");
        return RL_E_INVALID;
    }
    const std::vector<int> distinct = plan_ranks(device, devices, n_devices, rank_device, leader, comm_rank);
    *n_communicator_ranks = distinct.size() > 1 ? (uint32_t)distinct.size() : 0u;
    return RL_OK;
}
