// Randomised stress test of the inference scheduler + paged block manager, meant to be built with
// -fsanitize=address,undefined (host code only): thousands of requests with shared prefixes, random stop
// tokens and stop sequences, prompts longer than the prefill budget (chunked prefill), decode plans chained
// through continuation plans (results processed while the successor is "in flight"), pauses, aborts and
// cache flushes on a small page pool, so that preemption, prefix-cache reuse, deferred page release and
// page recycling all happen. Invariants checked along the way and at the end of every round:
//   * a chunked prompt is prefilled exactly once, in order, and only its last chunk samples,
//   * pages of a request that finished while a successor plan was in flight are handed to nobody before
//     that successor has been processed,
//   * every request finishes exactly once, with 1..max_new_tokens generated tokens unless aborted,
//   * log-probs / versions have one entry per generated token,
//   * no page leaks: with nothing running every page is free again (unreferenced prefix-cache pages
//     count as free: they are evictable).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>

#include "scheduler.hpp"

using namespace areal;

#define CHECK(cond, ...)                                   \
  do {                                                     \
    if (!(cond)) {                                         \
      std::fprintf(stderr, "CHECK failed: %s | ", #cond);  \
      std::fprintf(stderr, __VA_ARGS__);                   \
      std::fprintf(stderr, "\n");                          \
      std::exit(1);                                        \
    }                                                      \
  } while (0)

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 6;
  std::mt19937 rng(1234);
  long total_finished = 0, total_aborted = 0, total_preempt = 0, total_chunks = 0, total_continuations = 0,
       total_deferred = 0, total_seq_stops = 0;
  for (int round = 0; round < rounds; ++round) {
    SchedulerConfig cfg;
    cfg.num_pages = 20 + 6 * (round % 3);  // tight: forces preemption
    cfg.max_running_requests = 8 + 4 * (round % 2);
    cfg.max_prefill_tokens = round % 2 ? 48 : 256;  // 48: most prompts need several chunks
    cfg.context_length = 400;
    cfg.decode_steps = 1 + round % 4;
    cfg.enable_prefix_cache = round % 2 == 0;
    cfg.policy = round % 3 == 0 ? "lpm" : "fcfs";
    cfg.schedule_conservativeness = 0.3f;  // admit optimistically so that decode runs out of pages
    Scheduler s(cfg);
    std::map<std::string, int> budget;
    std::set<std::string> done;
    std::map<int64_t, int> chunk_pos;  // request id -> end of its last prefill chunk
    std::vector<int32_t> shared(40);
    for (auto& t : shared) t = 2 + rng() % 500;
    const int n_req = 300;
    int submitted = 0, version = 0;
    long guard = 0;
    while ((int)done.size() < n_req) {
      CHECK(++guard < 2000000, "round %d does not terminate (%zu of %d done)", round, done.size(), n_req);
      // trickle requests in
      while (submitted < n_req && (rng() % 3 == 0 || !s.has_work())) {
        std::vector<int32_t> p;
        if (rng() % 2) p.assign(shared.begin(), shared.begin() + 8 + rng() % 32);
        const int extra = 1 + rng() % 60;
        for (int i = 0; i < extra; ++i) p.push_back(2 + rng() % 500);
        SamplingParams sp;
        sp.max_new_tokens = 1 + rng() % 64;
        sp.min_new_tokens = rng() % 4;
        sp.stop_token_ids = {1};
        if (rng() % 2) sp.stop_sequences = {{7, 7}, {9, 8, 7}};
        const std::string rid = "r" + std::to_string(round) + "-" + std::to_string(submitted++);
        budget[rid] = sp.max_new_tokens;
        s.add_request(rid, p, sp);
      }
      const unsigned dice = rng() % 200;
      if (dice == 0) { s.pause(); }
      if (dice == 1 || s.paused()) { if (rng() % 4 == 0) s.resume(); }
      if (dice == 2) { total_aborted += s.abort_all(); s.set_version(++version); s.flush_cache(); chunk_pos.clear(); }
      auto draw = [&]() -> int64_t {
        const unsigned d = rng() % 60;
        return d == 0 ? 1 : d < 6 ? 7 : d < 8 ? 8 : d < 10 ? 9 : 2 + rng() % 500;
      };
      Plan p = s.schedule();
      if (p.kind == 1) {
        std::vector<int64_t> tok(p.req_ids.size());
        std::vector<float> lp(p.req_ids.size(), -0.25f);
        for (auto& t : tok) t = draw();
        CHECK(p.cu_q.size() == p.req_ids.size() + 1, "cu_q shape");
        CHECK(p.sample.size() == p.req_ids.size() && p.chunk_end.size() == p.req_ids.size(), "chunk arrays");
        CHECK((int)p.input_ids.size() == p.cu_q.back(), "packed prefill tokens %zu vs %d", p.input_ids.size(), p.cu_q.back());
        CHECK((int)p.input_ids.size() <= cfg.max_prefill_tokens, "prefill plan of %zu tokens over the budget %d",
              p.input_ids.size(), cfg.max_prefill_tokens);
        for (auto slot : p.slot_mapping) CHECK(slot >= 0 && slot < (int64_t)cfg.num_pages * 32, "slot %ld out of range", (long)slot);
        for (size_t i = 0; i < p.req_ids.size(); ++i) {
          const int start = p.positions[p.cu_q[i]], len = p.cu_q[i + 1] - p.cu_q[i];
          CHECK(len > 0 && start + len == p.chunk_end[i], "chunk bounds");
          CHECK(p.cu_k[i + 1] - p.cu_k[i] == p.chunk_end[i], "context length of a chunk row");
          auto it = chunk_pos.find(p.req_ids[i]);
          if (it != chunk_pos.end()) CHECK(it->second == start, "chunk starts at %d, previous one ended at %d", start, it->second);
          if (p.sample[i]) chunk_pos.erase(p.req_ids[i]); else { chunk_pos[p.req_ids[i]] = p.chunk_end[i]; ++total_chunks; }
          if (!p.sample[i]) CHECK(p.penalty_slot[i] == -1, "a non-sampling chunk must not touch the count table");
        }
        if (p.has_prefix) {
          CHECK(p.max_pages * 32 >= p.max_k && p.block_tables.size() == p.req_ids.size() * (size_t)p.max_pages,
                "page table of a prefill plan: %zu entries for %zu rows of %d pages", p.block_tables.size(),
                p.req_ids.size(), p.max_pages);
          for (size_t i = 0; i < p.req_ids.size(); ++i) {
            // the slots the plan writes lie in the pages its table names
            for (int t = p.cu_q[i]; t < p.cu_q[i + 1]; ++t)
              CHECK(p.slot_mapping[t] / 32 == p.block_tables[i * p.max_pages + p.positions[t] / 32], "slot outside its page");
          }
        }
        s.process_prefill(p, tok.data(), lp.data());
      } else if (p.kind == 2) {
        // a chain of decode plans: each successor is scheduled before its predecessor is processed
        Plan cur = std::move(p);
        std::set<int> quarantine;                    // pages of rows that finished with a successor in flight
        std::map<int64_t, size_t> pages_before;      // request -> pages it owned in the previous plan
        for (int depth = 0;; ++depth) {
          const size_t B = cur.req_ids.size();
          CHECK(cur.block_tables.size() == B * (size_t)cur.max_pages, "block table shape");
          for (auto pg : cur.block_tables) CHECK(pg >= 0 && pg < cfg.num_pages, "page %d out of range", pg);
          Plan nxt;
          if (depth < 6 && rng() % 3) nxt = s.schedule_continuation(cur, 1);
          const bool chained = nxt.kind == 2;
          if (chained) {
            CHECK(nxt.continuation && nxt.req_ids.size() == B && nxt.kill.size() == B, "continuation shape");
            ++total_continuations;
            // pages newly handed to live rows must not be ones the device may still write for dead rows
            for (size_t row = 0; row < B; ++row) {
              if (nxt.req_ids[row] < 0) { CHECK(nxt.kill[row] == 1, "dead row not killed"); continue; }
              CHECK(nxt.kill[row] == 0 && nxt.req_ids[row] == cur.req_ids[row], "live row moved");
              size_t had = pages_before.count(nxt.req_ids[row]) ? pages_before[nxt.req_ids[row]] : 0;
              for (int j = 0; j < nxt.max_pages; ++j) {
                const int pg = nxt.block_tables[row * nxt.max_pages + j];
                if (pg != 0 && (size_t)j >= had) CHECK(!quarantine.count(pg), "page %d handed out while the device may still write to it", pg);
              }
            }
          }
          pages_before.clear();
          const Plan& widest = chained ? nxt : cur;
          for (size_t row = 0; row < B; ++row) {
            if (widest.req_ids[row] < 0) continue;
            size_t n = 0;
            for (int j = 0; j < widest.max_pages; ++j) n += widest.block_tables[row * widest.max_pages + j] != 0;
            pages_before[widest.req_ids[row]] = n;
          }
          std::vector<int64_t> tok(B * cur.n_steps);
          std::vector<float> lp(B * cur.n_steps, -0.5f);
          for (auto& t : tok) t = draw();
          s.process_decode(cur, tok.data(), lp.data(), (int)B, chained);
          quarantine.clear();  // leftovers of the plan before `cur` were released by this call
          if (chained) {
            for (int pg : s.deferred_pages()) quarantine.insert(pg);
            if (!quarantine.empty()) ++total_deferred;
          } else {
            CHECK(s.num_deferred_pages() == 0, "pages deferred although no successor is in flight");
            break;
          }
          cur = std::move(nxt);
        }
      }
      for (auto& f : s.take_finished()) {
        CHECK(done.insert(f.rid).second || f.finish_reason == "abort", "request %s finished twice", f.rid.c_str());
        CHECK(f.output_tokens.size() == f.output_logprobs.size() && f.output_tokens.size() == f.output_versions.size(),
              "per-token arrays of %s differ in length", f.rid.c_str());
        CHECK((int)f.output_tokens.size() <= budget[f.rid], "%s generated %zu > budget %d", f.rid.c_str(),
              f.output_tokens.size(), budget[f.rid]);
        if (f.finish_reason == "stop") {
          const auto& o = f.output_tokens;
          const size_t n = o.size();
          const bool by_id = n >= 1 && o[n - 1] == 1;
          const bool by_seq = (n >= 2 && o[n - 1] == 7 && o[n - 2] == 7) || (n >= 3 && o[n - 3] == 9 && o[n - 2] == 8 && o[n - 1] == 7);
          CHECK(by_id || by_seq, "stop without stop token or stop sequence");
          if (!by_id) ++total_seq_stops;
        }
        if (f.finish_reason == "length") CHECK((int)f.output_tokens.size() == budget[f.rid], "length finish before the budget");
        for (auto v : f.output_versions) CHECK(v >= 0 && v <= version, "token version %d outside [0, %d]", v, version);
        ++total_finished;
      }
    }
    CHECK(!s.has_work(), "work left after all requests finished");
    CHECK(s.num_free_pages() == s.num_total_pages(), "page leak: free %d (cached %d) != total %d",
          s.num_free_pages(), s.num_cached_pages(), s.num_total_pages());
    s.flush_cache();
    CHECK(s.num_free_pages() == s.num_total_pages(), "pages still held after a cache flush");
    total_preempt += s.num_preemptions();
  }
  CHECK(total_chunks > 0 && total_continuations > 0 && total_deferred > 0 && total_seq_stops > 0,
        "a feature was never exercised: %ld chunks, %ld continuations, %ld deferred releases, %ld sequence stops",
        total_chunks, total_continuations, total_deferred, total_seq_stops);
  std::printf("stress ok: %ld finished, %ld aborted, %ld preemptions, %ld partial prefill chunks, %ld continuation plans, "
              "%ld deferred releases, %ld stop-sequence finishes\n", total_finished, total_aborted, total_preempt,
              total_chunks, total_continuations, total_deferred, total_seq_stops);
  return 0;
}
