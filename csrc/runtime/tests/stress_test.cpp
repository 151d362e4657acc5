// Randomised stress test of the inference scheduler + paged block manager, meant to be built with
// -fsanitize=address,undefined (host code only): thousands of requests with shared prefixes, random stop
// tokens, pauses, aborts and cache flushes on a small page pool, so that preemption, prefix-cache reuse
// and page recycling all happen. Invariants checked at the end of every round:
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
  long total_finished = 0, total_aborted = 0, total_preempt = 0;
  for (int round = 0; round < rounds; ++round) {
    SchedulerConfig cfg;
    cfg.num_pages = 20 + 6 * (round % 3);  // tight: forces preemption
    cfg.max_running_requests = 8 + 4 * (round % 2);
    cfg.max_prefill_tokens = 256;
    cfg.context_length = 400;
    cfg.decode_steps = 1 + round % 4;
    cfg.enable_prefix_cache = round % 2 == 0;
    cfg.policy = round % 3 == 0 ? "lpm" : "fcfs";
    cfg.schedule_conservativeness = 0.3f;  // admit optimistically so that decode runs out of pages
    Scheduler s(cfg);
    std::map<std::string, int> budget;
    std::set<std::string> done;
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
        const std::string rid = "r" + std::to_string(round) + "-" + std::to_string(submitted++);
        budget[rid] = sp.max_new_tokens;
        s.add_request(rid, p, sp);
      }
      const unsigned dice = rng() % 200;
      if (dice == 0) { s.pause(); }
      if (dice == 1 || s.paused()) { if (rng() % 4 == 0) s.resume(); }
      if (dice == 2) { total_aborted += s.abort_all(); s.set_version(++version); s.flush_cache(); }
      Plan p = s.schedule();
      if (p.kind == 1) {
        std::vector<int64_t> tok(p.req_ids.size());
        std::vector<float> lp(p.req_ids.size(), -0.25f);
        for (auto& t : tok) t = (rng() % 40 == 0) ? 1 : 2 + rng() % 500;
        CHECK(p.cu_q.size() == p.req_ids.size() + 1, "cu_q shape");
        CHECK((int)p.input_ids.size() == p.cu_q.back(), "packed prefill tokens %zu vs %d", p.input_ids.size(), p.cu_q.back());
        for (auto slot : p.slot_mapping) CHECK(slot >= 0 && slot < (int64_t)cfg.num_pages * 32, "slot %ld out of range", (long)slot);
        s.process_prefill(p, tok.data(), lp.data());
      } else if (p.kind == 2) {
        const size_t B = p.req_ids.size();
        CHECK(p.block_tables.size() == B * (size_t)p.max_pages, "block table shape");
        for (auto pg : p.block_tables) CHECK(pg >= 0 && pg < cfg.num_pages, "page %d out of range", pg);
        std::vector<int64_t> tok(B * p.n_steps);
        std::vector<float> lp(B * p.n_steps, -0.5f);
        for (auto& t : tok) t = (rng() % 40 == 0) ? 1 : 2 + rng() % 500;
        s.process_decode(p, tok.data(), lp.data(), (int)B);
      }
      for (auto& f : s.take_finished()) {
        CHECK(done.insert(f.rid).second || f.finish_reason == "abort", "request %s finished twice", f.rid.c_str());
        CHECK(f.output_tokens.size() == f.output_logprobs.size() && f.output_tokens.size() == f.output_versions.size(),
              "per-token arrays of %s differ in length", f.rid.c_str());
        CHECK((int)f.output_tokens.size() <= budget[f.rid], "%s generated %zu > budget %d", f.rid.c_str(),
              f.output_tokens.size(), budget[f.rid]);
        if (f.finish_reason == "stop") CHECK(!f.output_tokens.empty() && f.output_tokens.back() == 1, "stop without stop token");
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
  std::printf("stress ok: %ld finished, %ld aborted, %ld preemptions\n", total_finished, total_aborted, total_preempt);
  return 0;
}
