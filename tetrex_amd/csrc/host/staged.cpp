// Host side: the staged driver — a batch of queries is expanded piecewise (QueryExpansion,
// compiler.cpp), every piece is shipped as one blob to a StageExecutor (the device session), and the
// executor's answers about waiting states prune the frontier before it fans out (product code).
//
// One stage:
//   advance   every unfinished query emits ops until its budget is used (thread pool, largest first);
//             queries that no longer ask for feedback were already advanced while the previous stage
//             executed (`ahead`)
//   assemble  the stage's ops are ordered into dependency levels straight into the blob
//             (header | k-mer tables | programs | ops | levels, include/txq_program.h)
//   frontier  the waiting states of the queries that still ask for feedback
//   execute   StageExecutor::stage on a helper thread, overlapped with the next `ahead` expansion
//   prune     dead waiting states are dropped
#include "compiler.hpp"
#include "regex_front.hpp"
#include "thread_pool.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <stdexcept>
#include <sched.h>
#include <thread>

namespace tetrex {

namespace {

double now_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool env_is(const char* name, char value) {
    const char* e = std::getenv(name);
    return e && e[0] == value;
}

// CPUs this process may actually use: the hardware threads, its affinity mask, and the container's CPU quota
// (cgroup v2 cpu.max / v1 cpu.cfs_quota_us) — 8 ranks of a node that each start "all hardware threads" would
// oversubscribe it eight times
int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a > 0 && (n <= 0 || a < n)) n = a;
    }
    auto read_two = [](const char* path, double* a, double* b) {
        std::FILE* f = std::fopen(path, "r");
        if (!f) return 0;
        char x[64] = {0}, y[64] = {0};
        const int got = std::fscanf(f, "%63s %63s", x, y);
        std::fclose(f);
        if (got >= 1) *a = std::strcmp(x, "max") == 0 ? -1.0 : std::atof(x);
        if (got >= 2) *b = std::atof(y);
        return got;
    };
    double quota = -1, period = 100000;
    if (read_two("/sys/fs/cgroup/cpu.max", &quota, &period) < 1) {
        double q = -1, p = 100000, unused = 0;
        if (read_two("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &q, &unused) >= 1 && read_two("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &p, &unused) >= 1) { quota = q; period = p; }
    }
    if (quota > 0 && period > 0) {
        const int c = (int)(quota / period + 0.5);
        if (c >= 1 && (n <= 0 || c < n)) n = c;
    }
    return n < 1 ? 1 : n;
}

// default: this process's share of the CPUs it may use — all of them when it is alone on the node (the one-process
// deployment drives every GPU itself), 1 / LOCAL_WORLD_SIZE of them when a launcher started one rank per GPU beside it
// (torch.distributed.run and mpirun export the count) — at most 64: the expansion of 10 000 motifs stops gaining there
// (profiles/r3_host_scaling_10k_motifs.txt: 16 threads 21 ms, 32 threads 14 ms).
// Override with StagedOptions::threads or the TETREX_THREADS environment variable.
int expansion_threads(const StagedOptions& opt, size_t n_queries) {
    int threads = opt.threads;
    if (threads <= 0) {
        if (const char* env = std::getenv("TETREX_THREADS")) threads = std::atoi(env);
    }
    if (threads <= 0) {
        threads = usable_cpus();
        int ranks = 1;
        for (const char* name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS"})
            if (const char* env = std::getenv(name)) { ranks = std::atoi(env); break; }
        if (ranks > 1) threads = std::max(2, threads / ranks);
        if (threads > 64) threads = 64;
    }
    if (threads < 1) threads = 1;
    if ((size_t)threads > n_queries) threads = n_queries ? (int)n_queries : 1;
    return threads;
}

// The stage's blob is assembled in place, in storage that is reused from stage to stage.  The first
// large request reserves the most a stage can need: untouched pages cost nothing, and a buffer that
// never moves is faulted in once instead of once per growth step.
struct BlobStore {
    uint8_t* data = nullptr;
    size_t cap = 0;
    ~BlobStore() { std::free(data); }
    uint8_t* ensure(size_t bytes, size_t ceiling) {
        if (bytes > cap) {
            std::free(data);
            data = nullptr;
            cap = bytes > ((size_t)64 << 20) ? std::max(bytes + bytes / 2, ceiling) : bytes + bytes / 2;
            data = static_cast<uint8_t*>(BlockCache::fresh(cap));
        }
        return data;
    }
};

// Expansion threads are kept from run to run (spawning and joining 15 threads is half a millisecond of a 7 ms batch): a
// run borrows the idle pool of its size, or starts its own when that one is busy (runs on several indexes at once).
// With the threads come the buffers whose FIRST TOUCH is what costs: the stage blob (tens of MB for a 10 000-motif batch:
// thousands of page faults per run otherwise — 30 ms of a 100 ms batch) and the level scheduler's per-thread slot records.
struct RunBuffers {
    BlobStore blob;
    std::vector<LevelScratch> scratch;  // per thread
};
class PoolLease {
  public:
    explicit PoolLease(int threads) {
        {
            std::lock_guard<std::mutex> lk(mutex());
            for (Kept& k : kept())
                if (!k.busy && k.pool->threads() == threads) { k.busy = true; pool_ = k.pool.get(); buffers_ = k.buffers.get(); return; }
            if (kept().size() < 8) {
                kept().push_back(Kept{std::make_unique<ThreadPool>(threads), std::make_unique<RunBuffers>(), true});
                pool_ = kept().back().pool.get();
                buffers_ = kept().back().buffers.get();
                buffers_->scratch.resize((size_t)threads);
                return;
            }
        }
        own_ = std::make_unique<ThreadPool>(threads);
        own_buffers_ = std::make_unique<RunBuffers>();
        own_buffers_->scratch.resize((size_t)threads);
        pool_ = own_.get();
        buffers_ = own_buffers_.get();
    }
    ~PoolLease() {
        if (own_) return;
        std::lock_guard<std::mutex> lk(mutex());
        for (Kept& k : kept())
            if (k.pool.get() == pool_) k.busy = false;
    }
    PoolLease(const PoolLease&) = delete;
    PoolLease& operator=(const PoolLease&) = delete;
    ThreadPool& pool() { return *pool_; }
    RunBuffers& buffers() { return *buffers_; }

  private:
    struct Kept { std::unique_ptr<ThreadPool> pool; std::unique_ptr<RunBuffers> buffers; bool busy; };
    static std::mutex& mutex() { static std::mutex m; return m; }
    static std::vector<Kept>& kept() { static std::vector<Kept> v; return v; }
    ThreadPool* pool_ = nullptr;
    RunBuffers* buffers_ = nullptr;
    std::unique_ptr<ThreadPool> own_;
    std::unique_ptr<RunBuffers> own_buffers_;
};

class StagedRun {
  public:
    StagedRun(const KmerEncoder& enc, uint64_t bins, const std::vector<std::string>& regexes, StageExecutor& exec, const StagedOptions& opt)
        : enc_(enc), bins_(bins), regexes_(regexes), exec_(exec), opt_(opt), n_(regexes.size()), threads_(expansion_threads(opt, n_)), lease_(threads_), pool_(lease_.pool()),
          status_(n_, 0), why_(n_), q_(n_), passthrough_(n_, 0), ops_(n_), slots_(n_, TXQ_SLOT_FIRST_FREE), tables_(n_, KmerTable(false)),
          dgram_tables_(n_, KmerTable(false)), dense_ops_(n_), dslots_(n_, 0), tracked_(n_, 0), scratch_(lease_.buffers().scratch), dead_scratch_(threads_), levels_(n_), asks_(n_), fin_states_(n_, 0),
          fin_pruned_(n_, 0), ahead_(n_, 0), run_on_stages_(n_, 0), started_(n_, 0), unbuilt_(n_, 0), flushed_(n_, 0), released_(n_, 0), held_(n_, 0),
          busy_(threads_, 0.0), blob_store_(lease_.buffers().blob) {
        trace_ = std::getenv("TETREX_TRACE") != nullptr;  // per-stage phase times on stderr
        // unions of single residues as one k-graph node each (KGraph::kClass); TETREX_FUSE_CLASSES=0: the reference's node-for-node graph
        if (const char* e = std::getenv("TETREX_FUSE_CLASSES")) fuse_classes_ = std::atoi(e) != 0;
        if (const char* e = std::getenv("TETREX_LITERAL_FAST")) literal_fast_ = std::atoi(e) != 0;
        verified_levels_ = opt.verified_levels;
        if (std::getenv("TETREX_VERIFIED_LEVELS")) verified_levels_ = env_is("TETREX_VERIFIED_LEVELS", '1');  // A/B knob
        overlap_ = !env_is("TETREX_NO_OVERLAP", '1');
        run_on_budget_ = std::max(opt.ops_per_task, opt.ops_per_query_per_stage);
        dense_ = opt.dense;
        if (const char* e = std::getenv("TETREX_DENSE")) dense_.enabled = dense_.enabled && e[0] != '0';  // A/B knob: TETREX_DENSE=0
        if (const char* e = std::getenv("TETREX_DENSE_MIN")) dense_.min_states = (uint32_t)std::max(1, std::atoi(e));
        if (const char* e = std::getenv("TETREX_DENSE_SPARSE_BELOW")) dense_.sparse_below = (uint32_t)std::max(0, std::atoi(e));
        if (const char* e = std::getenv("TETREX_DENSE_COOL")) dense_.cool_down = (uint32_t)std::max(0, std::atoi(e));
        if (const char* e = std::getenv("TETREX_DENSE_TRACKED")) dense_.tracked_force = e[0] == '0' ? -1 : 1;  // A/B knob and tests: 0 never, 1 every query
        if (opt.gaps.dgram_loaded || dense_block_slots(enc, dense_) == 0) dense_.enabled = false;  // version-2 blobs, as before
        uint64_t pool_bytes = opt.dense_pool_bytes;
        if (const char* e = std::getenv("TETREX_DENSE_POOL_MB")) pool_bytes = (uint64_t)std::max(0LL, std::atoll(e)) << 20;  // device memory for dense blocks
        dense_pool_.store((int64_t)std::min<uint64_t>(pool_bytes, (uint64_t)INT64_MAX));
        dense_total_ = dense_pool_.load();
        // (tracked blocks are laid out inside their lists' geometries — typically far smaller than A^(k-1) entries; a query that
        // does need more takes it from the pool like everybody else)
        admit_bytes_ = (int64_t)std::min<uint64_t>(4 * dense_block_slots(enc, dense_) * (dense_.slot_bytes ? dense_.slot_bytes : 128), (uint64_t)256 << 20);
        dense_.pool = &dense_pool_;
        dense_.pool_total = dense_total_;
        evidence_.store(opt.dense_evidence);
        if (const char* e = std::getenv("TETREX_DENSE_EVIDENCE"))  // A/B knob and tests: dense / sparse / ask
            evidence_.store(e[0] == 'd' ? DenseOptions::kDense : e[0] == 's' ? DenseOptions::kSparse : e[0] == 't' ? DenseOptions::kThin : DenseOptions::kUnknown);
        dense_.evidence = &evidence_;
        wave_ops_ = opt.wave_ops;
        if (const char* e = std::getenv("TETREX_WAVE_OPS")) wave_ops_ = (size_t)std::max(0LL, std::atoll(e));  // A/B knob; 0 = one wave
        if (const char* e = std::getenv("TETREX_WAVE_GROWTH")) wave_growth_percent_ = (size_t)std::max(0LL, std::atoll(e));  // A/B knob: a wave is at least this share (%) of all ops emitted before it
        if (const char* e = std::getenv("TETREX_TASK_OPS")) run_on_budget_ = std::max<size_t>((size_t)std::atoll(e), 1);  // A/B knob
    }

    StagedStats run(std::vector<int>* status, std::vector<std::string>* messages) {
        lap_at_ = now_seconds();
        build_expansions();
        lap("graphs");
        double mark = now_seconds();
        for (bool first = true;; first = false) {
            const size_t stage_ops = advance_stage(first);
            bool pending = false;
            for (size_t i = 0; i < n_; ++i) pending |= unbuilt_[i] || (q_[i] && !q_[i]->done());
            if (!first && stage_ops == 0 && !pending) break;
            const Blob blob = assemble();
            Frontier fr = collect_frontier();
            st_.expand_seconds += now_seconds() - mark;
            mark = now_seconds();
            execute(blob, fr);
            mark = now_seconds();
            ++st_.stages;
            st_.ops += stage_ops;
            st_.kmers += blob.kmers;
            st_.feedback_queries += fr.program.size();
            prune(fr);
            if (!pending) break;
        }
        for (size_t i = 0; i < n_; ++i) {
            if (q_[i]) { st_.states += q_[i]->states(); st_.pruned += q_[i]->pruned(); }
            else { st_.states += fin_states_[i]; st_.pruned += fin_pruned_[i]; }
            st_.tracked_queries += tracked_[i];
        }
        if (status) *status = status_;
        if (messages) *messages = why_;
        st_.dense_evidence = evidence_.load();
        return st_;
    }

  private:
    struct Blob { const uint8_t* data; size_t bytes; size_t kmers; };
    struct Frontier {
        std::vector<uint32_t> queries;         // the queries that asked, in the order of the questions
        std::vector<size_t> first;             // [queries.size() + 1] range of each query's questions
        std::vector<uint32_t> program, slot;   // the questions: is slot `slot[a]` of program `program[a]` alive?
        std::vector<uint8_t> alive;            // the answers
        std::vector<uint32_t> run_on;          // unfinished queries that did not ask
        bool waiting = false;                  // there are queries that have not begun (a later wave)
    };

    void lap(const char* what) {
        if (!trace_) return;
        const double now = now_seconds();
        std::fprintf(stderr, "[tetrex] %-10s %8.2f ms\n", what, (now - lap_at_) * 1e3);
        lap_at_ = now;
    }

    // A query's k-graph and expansion are built when the query begins (advance_set), i.e. wave by wave: only the first
    // wave's graphs are built before the device has something to do.
    void build_expansions() {
        for (size_t i = 0; i < n_; ++i) {
            if (bins_ <= 1) passthrough_[i] = 1;  // include/query.h:265-272
            else { unbuilt_[i] = 1; pending_.push_back((uint32_t)i); }
        }
        // the longest first (a stage ends when its last task ends; nothing else is known about a query that has not been built)
        std::stable_sort(pending_.begin(), pending_.end(), [&](uint32_t x, uint32_t y) { return regexes_[x].size() > regexes_[y].size(); });
    }
    void build_one(size_t i) {  // throws what the front-end throws
        std::string plain;
        const std::string postfix = preprocess_query(regexes_[i], enc_, &plain);
        // a plain string of residues needs no graph (QueryExpansion's literal constructor); TETREX_LITERAL_FAST=0: the general way
        if (literal_fast_ && enc_.alphabet() == Alphabet::Base && !plain.empty() &&
            std::all_of(plain.begin(), plain.end(), [](char c) { return c >= 'A' && c <= 'Z'; })) {
            q_[i] = std::make_unique<QueryExpansion>(enc_, std::move(plain), opt_.limits);
            return;
        }
        q_[i] = std::make_unique<QueryExpansion>(enc_, build_kgraph(postfix, enc_.k(), enc_.alphabet() != Alphabet::Base, opt_.gaps.augment, fuse_classes_), opt_.limits, opt_.gaps, dense_);
    }

    // ops of earlier stages only ever reach RESULT through a Match op, so an abandoned query is
    // neutralised by not emitting anything further
    void fail(size_t i, const std::exception& e) {
        ops_[i].clear();
        tables_[i].clear();
        dgram_tables_[i].clear();
        dense_ops_[i].clear();
        if (q_[i]) held_[i] = q_[i]->pool_taken();
        q_[i].reset();
        flushed_[i] = 1;  // nothing of it is left to run
        status_[i] = -1;
        why_[i] = e.what();
    }

    // Expands the queries under way (`active`, largest first) until each has used its budget, and lets queries that have not
    // begun (`pending_`, longest first) begin while the wave and the dense pool have room; returns the ops emitted.  The
    // workers stop taking pending queries at the first one that may not begin: a batch of 10 000 queries does not pay one
    // contended counter increment per waiting query and stage (that, and sorting all of them, was a third of the host's time).
    size_t advance_set(std::vector<uint32_t>& active, size_t already, size_t feedback_budget, std::vector<uint32_t>* begun_now = nullptr) {
        {
            std::vector<std::pair<uint64_t, uint32_t>> w(active.size());
            for (size_t j = 0; j < active.size(); ++j) w[j] = {q_[active[j]] ? q_[active[j]]->weight() : 0, active[j]};
            std::stable_sort(w.begin(), w.end(), [](const auto& x, const auto& y) { return x.first > y.first; });  // a stage ends when its last task ends
            for (size_t j = 0; j < active.size(); ++j) active[j] = w[j].second;
        }
        std::fill(busy_.begin(), busy_.end(), 0.0);
        std::atomic<size_t> total{already};
        // Queries begin in waves: the first stage goes to the device after about wave_ops_ ops, and the later waves are
        // expanded while it executes (execute()); the waves grow with what has been emitted, so a large batch does not
        // turn into many small stages.
        const size_t wave_limit = wave_ops_ ? std::max<size_t>(wave_ops_, (size_t)(st_.ops * wave_growth_percent_ / 100)) : 0;
        // Dense blocks are device memory: when the run's pool runs low, queries that have not begun wait for a later stage —
        // those under way finish, hand their blocks back (and the device recycles their regions) — instead of everybody
        // starting at once and the late ones falling back to enumerated states.  Somebody is always under way.
        size_t under_way = 0;
        for (uint32_t i : active) under_way += q_[i] && !q_[i]->done();
        std::atomic<size_t> begun{under_way};
        // returns false when query i has not begun and may not begin now
        auto task = [&](size_t i, int t) -> bool {
            const double t0 = trace_ ? now_seconds() : 0.0;
            if (total.load(std::memory_order_relaxed) >= opt_.ops_per_stage) return false;  // waits for a later stage
            if (!started_[i]) {
                if (wave_limit && total.load(std::memory_order_relaxed) >= wave_limit && begun.load(std::memory_order_relaxed) > 0) return false;
                // admission by an estimate of four blocks per query (what a chain of steps through x(m,n) gaps holds at a time)
                if (dense_.enabled && admitted_.fetch_add(admit_bytes_, std::memory_order_relaxed) + admit_bytes_ > dense_total_ &&
                    begun.load(std::memory_order_relaxed) > 0) {
                    admitted_.fetch_sub(admit_bytes_, std::memory_order_relaxed);
                    return false;
                }
                started_[i] = 1;
                begun.fetch_add(1, std::memory_order_relaxed);
                if (unbuilt_[i]) {
                    unbuilt_[i] = 0;
                    try {
                        build_one(i);
                    } catch (const std::exception& e) {
                        fail(i, e);
                        return true;
                    }
                }
            }
            try {
                // a query that gains nothing from feedback only pauses to keep the stage's tasks even
                const bool asks = q_[i]->wants_feedback();
                // a feedback-free query that keeps coming back quadruples its budget each time (256 k, 1 M, 4 M)
                const size_t grown = run_on_budget_ << (2 * std::min<uint32_t>(run_on_stages_[i], 2));
                if (!asks) ++run_on_stages_[i];
                q_[i]->advance(asks ? feedback_budget : grown, tables_[i], ops_[i], &dgram_tables_[i],
                               asks && verified_levels_ && q_[i]->mostly_dying(), dense_.enabled ? &dense_ops_[i] : nullptr);
            } catch (const std::exception& e) {
                fail(i, e);
                return true;
            }
            total.fetch_add(ops_[i].size(), std::memory_order_relaxed);
            slots_[i] = q_[i]->n_slots();
            dslots_[i] = q_[i]->n_dense_blocks();
            tracked_[i] = q_[i]->tracked();
            if (q_[i]->done()) {  // free the expansion's tables here, on the worker
                fin_states_[i] = q_[i]->states();
                fin_pruned_[i] = q_[i]->pruned();
                held_[i] = q_[i]->pool_taken();
                q_[i].reset();
            }
            if (trace_) busy_[t] += now_seconds() - t0;
            return true;
        };
        const size_t head = pending_head_;
        std::atomic<size_t> next_active{0}, next_pending{head};
        std::atomic<bool> closed{false};  // a pending query could not begin: none behind it is tried in this call
        pool_.run((size_t)threads_, [&](size_t, int t) {
            for (size_t at; (at = next_active.fetch_add(1, std::memory_order_relaxed)) < active.size();) task(active[at], t);
            while (!closed.load(std::memory_order_relaxed)) {
                const size_t at = next_pending.fetch_add(1, std::memory_order_relaxed);
                if (at >= pending_.size()) break;
                if (!task(pending_[at], t)) closed.store(true, std::memory_order_relaxed);
            }
        });
        // the pending queries taken in this call: those that began leave the list (the others stay in front, in their order)
        const size_t taken = std::min(next_pending.load(), pending_.size());
        auto mid = std::stable_partition(pending_.begin() + head, pending_.begin() + taken, [&](uint32_t i) { return started_[i] != 0; });
        if (begun_now) begun_now->assign(pending_.begin() + head, mid);
        pending_head_ = (size_t)(mid - pending_.begin());
        if (trace_) {
            double sum = 0, mx = 0;
            for (double b : busy_) { sum += b; if (b > mx) mx = b; }
            std::fprintf(stderr, "[tetrex] busy sum %8.2f ms max %8.2f ms ops %zu queries %zu + %zu that began\n", sum * 1e3, mx * 1e3, total.load(), active.size(),
                         pending_head_ - head);
        }
        return total.load() - already;
    }

    // advance: fills touched_ (queries with something for this stage) and returns the stage's op count
    size_t advance_stage(bool first) {
        // Queries whose last ops the device has been given in an earlier stage are through with their dense blocks: the
        // bytes go back to the pool and to the admission budget now, and this stage's program table reports them with zero
        // dense slots, so the device recycles their regions before it sizes those of the queries admitted next.
        if (dense_.enabled)
            for (size_t i = 0; i < n_; ++i)
                if (!q_[i] && started_[i] && flushed_[i] && !released_[i]) {
                    released_[i] = 1;
                    dense_pool_.fetch_add((int64_t)held_[i], std::memory_order_relaxed);
                    admitted_.fetch_sub(admit_bytes_, std::memory_order_relaxed);
                }
        touched_.clear();
        std::vector<uint32_t> act;
        size_t unfinished = 0;
        for (size_t i = 0; i < n_; ++i) {
            if (first && passthrough_[i]) {
                ops_[i].push_back(txq_op{TXQ_NO_KMER, TXQ_SLOT_RESULT, TXQ_SLOT_ONES, TXQ_SLOT_RESULT});
                touched_.push_back((uint32_t)i);
            }
            if (ahead_[i]) {  // expanded during the previous execution
                ahead_[i] = 0;
                if (!ops_[i].empty() || !tables_[i].values().empty()) touched_.push_back((uint32_t)i);
                unfinished += q_[i] && !q_[i]->done();
                continue;
            }
            if (unbuilt_[i]) ++unfinished;  // (has not begun: advance_set takes it from pending_)
            else if (q_[i] && !q_[i]->done()) { act.push_back((uint32_t)i); ++unfinished; }
        }
        // with few queries left, each gets a larger share of the stage (fewer, fuller stages)
        size_t feedback_budget = unfinished ? opt_.stage_target_ops / unfinished : opt_.ops_per_query_per_stage;
        if (feedback_budget < opt_.ops_per_query_per_stage) feedback_budget = opt_.ops_per_query_per_stage;
        if (feedback_budget > opt_.ops_per_task) feedback_budget = run_on_budget_;
        feedback_budget_ = feedback_budget;
        std::vector<uint32_t> begun_now;
        const size_t total = carried_ + advance_set(act, carried_, feedback_budget, &begun_now);
        carried_ = 0;
        for (const std::vector<uint32_t>* v : {&act, &begun_now})
            for (uint32_t i : *v)
                if (!ops_[i].empty() || !tables_[i].values().empty()) touched_.push_back(i);
        std::sort(touched_.begin(), touched_.end());
        lap("advance");
        return total;
    }

    // layout: header | k-mer tables of the touched queries, then their d-gram tables (the device
    // probes the last `stage_dgrams` entries on the auxiliary index) | programs | ops | levels
    Blob assemble() {
        const size_t m = touched_.size();
        std::vector<uint32_t> base(m), dbase(m), first_op(m), first_dense(m);
        size_t stage_kmers = 0, stage_dgrams = 0, stage_ops = 0, stage_dense = 0;
        for (size_t j = 0; j < m; ++j) {
            base[j] = (uint32_t)stage_kmers;
            stage_kmers += tables_[touched_[j]].values().size();
            first_op[j] = (uint32_t)stage_ops;
            stage_ops += ops_[touched_[j]].size();
            first_dense[j] = (uint32_t)stage_dense;
            stage_dense += dense_ops_[touched_[j]].size();
        }
        st_.dense_ops += stage_dense;
        for (size_t j = 0; j < m; ++j) {
            dbase[j] = (uint32_t)(stage_kmers + stage_dgrams);
            stage_dgrams += dgram_tables_[touched_[j]].values().size();
        }
        if (stage_kmers + stage_dgrams > 0x7FFFFFF0u) throw std::runtime_error("stage k-mer table overflow");
        if (stage_ops > 0xFFFFFFFFu) throw std::runtime_error("stage has more than 2^32 operations");
        // version 3 (dense table between the programs and the ops) whenever this run may use dense blocks
        const bool v3 = dense_.enabled;
        txq_blob_header_v3 h3{};
        txq_blob_header_v2& h = h3.v2;
        h.magic = TXQ_PROGRAM_MAGIC;
        h.version = v3 ? TXQ_PROGRAM_VERSION_DENSE : TXQ_PROGRAM_VERSION_LEVELS;
        h.n_programs = (uint32_t)n_;
        h.n_kmers = (uint32_t)(stage_kmers + stage_dgrams);
        h.n_ops = (uint32_t)stage_ops;
        h.n_aux_kmers = stage_dgrams;
        h.kmers_offset = v3 ? sizeof(txq_blob_header_v3) : sizeof(txq_blob_header_v2);
        h.programs_offset = h.kmers_offset + (stage_kmers + stage_dgrams) * sizeof(uint64_t);
        h3.dense_offset = h.programs_offset + n_ * sizeof(txq_program_v2);
        h3.n_dense = (uint32_t)stage_dense;
        h3.k = enc_.k();
        h3.bits = enc_.bits_per_symbol();
        h3.alphabet = enc_.alphabet_size();
        h3.canonical = enc_.molecule() == Molecule::DNA ? 1u : 0u;
        h.ops_offset = h3.dense_offset + (v3 ? stage_dense * sizeof(txq_dense_op) : 0);
        h.levels_offset = h.ops_offset + stage_ops * sizeof(txq_op);
        // a program has at most one level per op, and per op at worst one k-mer: the ceiling of the reservation
        const size_t most_ops = opt_.ops_per_stage + (size_t)threads_ * std::min<size_t>(run_on_budget_ << 4, std::min<size_t>(opt_.limits.max_ops, (size_t)64 << 20));
        uint8_t* blob = blob_store_.ensure(h.levels_offset + stage_ops * 4 + 8,
                                           sizeof(txq_blob_header_v3) + n_ * sizeof(txq_program_v2) + most_ops * (sizeof(txq_op) + 4 + 8));
        txq_dense_op* blob_dense = reinterpret_cast<txq_dense_op*>(blob + h3.dense_offset);
        uint64_t* blob_kmers = reinterpret_cast<uint64_t*>(blob + h.kmers_offset);
        txq_op* blob_ops = reinterpret_cast<txq_op*>(blob + h.ops_offset);
        pool_.run(m, [&](size_t j, int t) {
            const uint32_t i = touched_[j];
            const KmerVec& km = tables_[i].values();
            if (!km.empty()) std::memcpy(blob_kmers + base[j], km.data(), km.size() * 8);
            const KmerVec& dg = dgram_tables_[i].values();
            if (!dg.empty()) std::memcpy(blob_kmers + dbase[j], dg.data(), dg.size() * 8);
            const DenseVec& dn = dense_ops_[i];
            if (!dn.empty()) std::memcpy(blob_dense + first_dense[j], dn.data(), dn.size() * sizeof(txq_dense_op));
            const DenseSchedule ds{dn.data(), first_dense[j], dslots_[i]};
            levels_[i] = schedule_levels_into(ops_[i], slots_[i], scratch_[t], blob_ops + first_op[j], base[j], dbase[j], v3 ? &ds : nullptr);
        });
        lap("levels");
        txq_program_v2* pr = reinterpret_cast<txq_program_v2*>(blob + h.programs_offset);
        uint32_t* lv = reinterpret_cast<uint32_t*>(blob + h.levels_offset);
        size_t j = 0;
        uint32_t at_level = 0;
        for (size_t i = 0; i < n_; ++i) {
            if (j < m && touched_[j] == i) {
                const uint32_t nl = (uint32_t)levels_[i].size();
                pr[i] = txq_program_v2{first_op[j], (uint32_t)ops_[i].size(), slots_[i], at_level, nl, dslots_[i] | (tracked_[i] ? TXQ_PROGRAM_TRACKED_BIT : 0u)};
                if (nl) std::memcpy(lv + at_level, levels_[i].data(), (size_t)nl * 4);
                at_level += nl;
                ++j;
            } else {
                if (released_[i]) dslots_[i] = 0;  // its dense region goes to the queries admitted now (see advance_stage)
                pr[i] = txq_program_v2{(uint32_t)stage_ops, 0, slots_[i], at_level, 0, dslots_[i] | (tracked_[i] ? TXQ_PROGRAM_TRACKED_BIT : 0u)};
            }
        }
        h.n_levels = at_level;
        if (at_level & 1) lv[at_level] = 0;
        if (v3) std::memcpy(blob, &h3, sizeof h3);
        else std::memcpy(blob, &h, sizeof h);
        lap("programs");
        // the blob holds the stage now: the per-query buffers are free for the next one
        // A query that goes on has its buffers emptied here, before it emits again.  A FINISHED query's buffers go back to the
        // allocator — thousands of frees per stage of a large batch, by threads that did not allocate them — once the stage has
        // been handed to the device (release_finished, from execute()): nothing waits for them.
        pool_.run(touched_.size(), [&](size_t j, int) {
            const uint32_t i = touched_[j];
            dense_ops_[i].clear();
            if (q_[i]) { ops_[i].clear(); tables_[i].clear(); dgram_tables_[i].clear(); }
            else flushed_[i] = 1;  // finished, and its last ops are in this blob
            levels_[i].clear();
        });
        for (const uint32_t i : touched_)
            if (!q_[i]) finished_.push_back(i);
        lap("blob");
        return Blob{blob, h.levels_offset + (((size_t)h.n_levels * 4 + 7) & ~(size_t)7), stage_kmers + stage_dgrams};
    }

    // which waiting states does the device have to report on
    Frontier collect_frontier() {
        Frontier fr;
        for (size_t i = 0; i < n_; ++i) {
            if (unbuilt_[i]) { fr.waiting = true; continue; }
            if (!q_[i] || q_[i]->done()) continue;
            (q_[i]->wants_feedback() ? fr.queries : fr.run_on).push_back((uint32_t)i);
        }
        pool_.run(fr.queries.size(), [&](size_t j, int) {
            const uint32_t i = fr.queries[j];
            asks_[i].clear();
            q_[i]->frontier_slots(asks_[i]);
        });
        fr.first.assign(fr.queries.size() + 1, 0);
        for (size_t j = 0; j < fr.queries.size(); ++j) {
            const std::vector<uint32_t>& v = asks_[fr.queries[j]];
            fr.slot.insert(fr.slot.end(), v.begin(), v.end());
            fr.program.insert(fr.program.end(), v.size(), fr.queries[j]);
            fr.first[j + 1] = fr.slot.size();
        }
        fr.alive.assign(fr.program.size(), 1);
        lap("frontier");
        return fr;
    }

    // The device runs the stage; meanwhile the queries that do not wait for its answer go on, and the next wave of
    // queries begins (`ahead`; their ops are carried into the next stage).
    void release_finished() {
        pool_.run(finished_.size(), [&](size_t j, int) {
            const uint32_t i = finished_[j];
            OpVec().swap(ops_[i]);
            tables_[i] = KmerTable(false);
            dgram_tables_[i] = KmerTable(false);
            DenseVec().swap(dense_ops_[i]);
        });
        finished_.clear();
        lap("release");
    }

    void execute(const Blob& blob, Frontier& fr) {
        const double start = now_seconds();
        if (overlap_ && (!fr.run_on.empty() || fr.waiting)) {
            std::future<void> running = std::async(std::launch::async, [&]() { exec_.stage(blob.data, blob.bytes, fr.program, fr.slot, fr.alive); });
            std::vector<uint32_t> set = fr.run_on, begun_now;
            try {
                carried_ = advance_set(set, 0, feedback_budget_, &begun_now);
            } catch (...) {
                running.wait();
                throw;
            }
            for (uint32_t i : set) ahead_[i] = 1;
            for (uint32_t i : begun_now) ahead_[i] = 1;
            const double ahead_s = now_seconds() - start;
            lap("ahead");
            release_finished();
            running.get();
            st_.expand_seconds += ahead_s;
            st_.execute_seconds += now_seconds() - start - ahead_s;  // what the device added beyond the overlapped expansion
        } else {
            if (overlap_ && !finished_.empty()) {  // (nothing to expand beside the stage: the buffers of the finished queries still are)
                std::future<void> running = std::async(std::launch::async, [&]() { exec_.stage(blob.data, blob.bytes, fr.program, fr.slot, fr.alive); });
                release_finished();
                running.get();
            } else {
                exec_.stage(blob.data, blob.bytes, fr.program, fr.slot, fr.alive);
                release_finished();
            }
            st_.execute_seconds += now_seconds() - start;
        }
        lap("execute");
    }

    void prune(const Frontier& fr) {
        // while nobody knows how states fare on this index, the answers are also read as mask fills (see DenseOptions::evidence)
        const bool look = dense_.enabled && evidence_.load(std::memory_order_relaxed) == DenseOptions::kUnknown;
        std::atomic<uint64_t> seen_bits{0}, seen_states{0};
        pool_.run(fr.queries.size(), [&](size_t j, int t) {
            const uint32_t p = fr.queries[j];
            if (look && q_[p]) {
                std::vector<uint8_t>& klass = dead_scratch_[t];
                klass.assign(q_[p]->n_slots(), 0xFF);
                for (size_t a = fr.first[j]; a < fr.first[j + 1]; ++a) klass[fr.slot[a]] = fr.alive[a];
                uint64_t b = 0, n = 0;
                q_[p]->observe(klass, &b, &n);
                seen_bits.fetch_add(b, std::memory_order_relaxed);
                seen_states.fetch_add(n, std::memory_order_relaxed);
            }
            bool any = false;
            for (size_t a = fr.first[j]; a < fr.first[j + 1] && !any; ++a) any = !fr.alive[a];
            if (!any) return;
            std::vector<uint8_t>& dead = dead_scratch_[t];
            dead.assign(q_[p]->n_slots(), 0);
            for (size_t a = fr.first[j]; a < fr.first[j + 1]; ++a)
                if (!fr.alive[a]) dead[fr.slot[a]] = 1;
            q_[p]->prune(dead);
        });
        if (look && seen_states.load() >= 16) {  // a handful of states is no basis: the next stage looks again
            const double fill = std::min(1.0, (double)seen_bits.load() / ((double)seen_states.load() * (double)std::max<uint64_t>(opt_.feedback_bins ? opt_.feedback_bins : bins_, 1)));
            st_.observed_fill = fill;
            const double bits_per_state = (double)seen_bits.load() / (double)seen_states.load();
            const int verdict = fill >= opt_.dense_min_fill ? DenseOptions::kDense : bits_per_state < opt_.thin_bits ? DenseOptions::kThin : DenseOptions::kSparse;
            evidence_.store(verdict);
            if (trace_) std::fprintf(stderr, "[tetrex] masks of %llu probed states are %.1f %% full (%.1f bits): %s\n", (unsigned long long)seen_states.load(), fill * 100,
                                     bits_per_state, verdict == DenseOptions::kDense ? "lists saturate, dense steps" : verdict == DenseOptions::kThin ? "states die out, tracked blocks where the executor keeps lists" : "states thin out, fewer blocks");
        }
        lap("prune");
    }

    const KmerEncoder& enc_;
    const uint64_t bins_;
    const std::vector<std::string>& regexes_;
    StageExecutor& exec_;
    const StagedOptions& opt_;
    const size_t n_;
    const int threads_;
    PoolLease lease_;
    ThreadPool& pool_;  // every query is expanded by one thread at a time; threads own disjoint queries

    std::vector<int> status_;
    std::vector<std::string> why_;
    std::vector<std::unique_ptr<QueryExpansion>> q_;  // null: failed, finished or a pass-through
    std::vector<uint8_t> passthrough_;
    std::vector<uint32_t> finished_;  // queries whose last ops have gone into a blob: their buffers are released beside the stage (release_finished)
    std::vector<OpVec> ops_;       // per query: the ops of the stage being built
    std::vector<uint32_t> slots_;  // per query: slots in use (what the device sizes the slot region by)
    // one k-mer table per query and stage: small enough to stay cache-resident, and a k-mer shared
    // by two queries is simply probed twice (a probe costs far less than a shared-table miss)
    std::vector<KmerTable> tables_, dgram_tables_;
    std::vector<DenseVec> dense_ops_;  // per query: the dense ops of the stage being built (op.dst of a TXQ_DENSE_OP indexes it)
    std::vector<uint32_t> dslots_;     // per query: dense blocks it addresses (ids 0 .. n-1)
    std::vector<uint8_t> tracked_;     // per query: its blocks carry live lists (TXQ_PROGRAM_TRACKED_BIT)
    DenseOptions dense_;
    std::atomic<int64_t> dense_pool_{0};
    std::atomic<int> evidence_{DenseOptions::kUnknown};  // see DenseOptions::evidence
    std::vector<LevelScratch>& scratch_;              // per thread (kept with the thread pool from run to run)
    std::vector<std::vector<uint8_t>> dead_scratch_;  // per thread
    std::vector<std::vector<uint32_t>> levels_, asks_;
    std::vector<uint64_t> fin_states_, fin_pruned_;   // statistics of the queries freed early
    std::vector<uint32_t> touched_;                   // queries with ops or k-mers in the stage being built, ascending
    std::vector<uint8_t> ahead_;                      // advanced while the previous stage executed
    std::vector<uint32_t> run_on_stages_;             // per query: stages it has run without asking for feedback
    std::vector<uint8_t> started_;                    // per query: its expansion has begun (waves, dense-memory admission)
    std::vector<uint32_t> pending_;                   // the queries that have not begun are pending_[pending_head_ ..], longest first
    size_t pending_head_ = 0;
    std::vector<uint8_t> unbuilt_;                    // per query: its k-graph is still to be built (when it begins)
    std::vector<uint8_t> flushed_, released_;         // finished and its last ops handed to the device / its blocks handed back
    std::vector<uint64_t> held_;                      // bytes of the dense pool a finished query still holds
    int64_t dense_total_ = 0, admit_bytes_ = 0;       // the pool's size; what a query is assumed to need when it is admitted
    std::atomic<int64_t> admitted_{0};
    size_t carried_ = 0;                              // ops those produced
    std::vector<double> busy_;
    BlobStore& blob_store_;                           // (kept with the thread pool from run to run)
    StagedStats st_;
    size_t run_on_budget_ = 0, feedback_budget_ = 0;  // the latter: what a query that asks gets per stage (advance_stage sets it)
    size_t wave_ops_ = 0, wave_growth_percent_ = 100;
    bool trace_ = false, verified_levels_ = true, overlap_ = true, fuse_classes_ = true, literal_fast_ = true;
    double lap_at_ = 0;
};

}  // namespace

std::vector<uint64_t> join_shard_masks(size_t n, uint64_t mask_words, const std::vector<uint64_t>& word0, const std::vector<uint64_t>& words,
                                       const std::vector<const uint64_t*>& shard_masks) {
    if (word0.size() != words.size() || word0.size() != shard_masks.size()) throw std::runtime_error("shard tables of different lengths");
    std::vector<uint8_t> covered(mask_words, 0);
    for (size_t r = 0; r < word0.size(); ++r) {
        if (word0[r] > mask_words || words[r] > mask_words - word0[r]) throw std::runtime_error("shard outside the mask");
        for (uint64_t w = 0; w < words[r]; ++w) {
            if (covered[word0[r] + w]) throw std::runtime_error("shards overlap");
            covered[word0[r] + w] = 1;
        }
    }
    for (uint8_t c : covered)
        if (!c) throw std::runtime_error("shards do not cover the mask");
    std::vector<uint64_t> full(n * mask_words);
    for (size_t r = 0; r < word0.size(); ++r)
        for (size_t i = 0; i < n && words[r]; ++i)
            std::memcpy(full.data() + i * mask_words + word0[r], shard_masks[r] + i * words[r], words[r] * 8);
    return full;
}

StagedStats run_staged(const KmerEncoder& enc, uint64_t bins, const std::vector<std::string>& regexes, StageExecutor& exec,
                       const StagedOptions& opt, std::vector<int>* status, std::vector<std::string>* messages) {
    return StagedRun(enc, bins, regexes, exec, opt).run(status, messages);
}

}  // namespace tetrex
