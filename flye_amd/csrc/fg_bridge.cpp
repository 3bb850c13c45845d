// Host-side batch scheduler above the C ABI (include/flye_gpu_bridge.h):
// many caller threads, one dispatcher thread that owns the fg_ctx.
//
// Mirrors the query side of the reference's OverlapContainer (src/sequence/overlap.cpp:
// 518-574): lazySeqOverlaps = cached forward list + complemented twin, quickSeqOverlaps =
// uncached, with the caller's maxOverlaps / forceLocal.
#include "../../include/flye_gpu_bridge.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

typedef std::vector<fg_overlap_rec> RecList;

// OverlapRange::complement() (overlap.h:118-147)
fg_overlap_rec complement(const fg_overlap_rec& o)
{
	fg_overlap_rec r = o;
	r.cur_begin = o.cur_len - o.cur_end - 1;
	r.cur_end = o.cur_len - o.cur_begin - 1;
	r.ext_begin = o.ext_len - o.ext_end - 1;
	r.ext_end = o.ext_len - o.ext_begin - 1;
	r.cur_id = o.cur_id ^ 1u;
	r.ext_id = o.ext_id ^ 1u;
	return r;
}

struct Entry {	// one forward read of the lazy cache (IndexVecWrapper, overlap.h:401-416)
	bool ready = false;
	int status = FG_OK;
	RecList fwd, rev;
};

struct QuickReq {
	uint32_t id; int32_t maxOverlaps; uint8_t forceLocal;
	bool done = false; int status = FG_OK;
	RecList out;
};

} // namespace

struct fgb_container {
	fg_ctx* ctx = nullptr;
	fg_detector_params params;
	uint32_t maxBatch = 4096, lingerUs = 200;

	std::mutex mu;
	std::condition_variable cvWork, cvDone;
	std::unordered_map<uint32_t, std::unique_ptr<Entry>> cache;	// by forward id
	std::deque<uint32_t> lazyQ;		// forward ids somebody waits for
	std::deque<uint32_t> prefetchQ;	// forward ids nobody waits for yet
	std::deque<QuickReq*> quickQ;
	std::vector<float> divStats;
	fgb_stats stats{0, 0, 0, 0, 0};
	bool stop = false;
	std::thread worker;

	void run();
	int deviceCall(const fg_detector_params& p, const std::vector<uint32_t>& ids, int32_t mo, uint8_t fl,
				   std::vector<RecList>& lists, std::vector<float>& st);
};

int fgb_container::deviceCall(const fg_detector_params& p, const std::vector<uint32_t>& ids, int32_t mo, uint8_t fl,
							  std::vector<RecList>& lists, std::vector<float>& st)
{
	fg_overlap_batch b;
	const int rc = fg_overlaps(ctx, &p, ids.data(), (uint32_t)ids.size(), mo, fl, &b);
	if (rc != FG_OK) return rc;
	lists.resize(ids.size());
	for (size_t i = 0; i < ids.size(); ++i)
		lists[i].assign(b.recs + b.query_off[i], b.recs + b.query_off[i + 1]);
	st.assign(b.div_stats, b.div_stats + b.n_div_stats);
	fg_release_batch(&b);
	return FG_OK;
}

void fgb_container::run()
{
	std::unique_lock<std::mutex> lk(mu);
	while (true)
	{
		cvWork.wait(lk, [&] { return stop || !lazyQ.empty() || !quickQ.empty() || !prefetchQ.empty(); });
		if (stop) break;
		// let the other worker threads of the caller pile their requests on
		if (lazyQ.size() + quickQ.size() < maxBatch && lingerUs)
			cvWork.wait_for(lk, std::chrono::microseconds(lingerUs),
							[&] { return stop || lazyQ.size() + quickQ.size() >= maxBatch; });
		if (stop) break;

		// ---- lazy class: maxOverlaps = 0, forceLocal = false (overlap.cpp:547-551) ----
		std::vector<uint32_t> ids;
		while (!lazyQ.empty() && ids.size() < maxBatch) { ids.push_back(lazyQ.front()); lazyQ.pop_front(); }
		while (!prefetchQ.empty() && ids.size() < maxBatch)
		{
			const uint32_t id = prefetchQ.front();
			prefetchQ.pop_front();
			auto it = cache.find(id);
			if (it != cache.end()) continue;	// asked for (and queued) or computed meanwhile
			cache.emplace(id, std::unique_ptr<Entry>(new Entry));
			ids.push_back(id);
		}
		if (!ids.empty())
		{
			const fg_detector_params p = params;
			lk.unlock();
			std::vector<RecList> lists; std::vector<float> st;
			const int rc = deviceCall(p, ids, 0, 0, lists, st);
			std::vector<RecList> revs(rc == FG_OK ? ids.size() : 0);
			for (size_t i = 0; i < revs.size(); ++i)
			{
				revs[i].reserve(lists[i].size());
				for (const auto& o : lists[i]) revs[i].push_back(complement(o));
			}
			lk.lock();
			++stats.device_calls; stats.reads_computed += ids.size();
			for (size_t i = 0; i < ids.size(); ++i)
			{
				Entry& e = *cache[ids[i]];
				e.status = rc;
				if (rc == FG_OK)
				{
					e.fwd.swap(lists[i]); e.rev.swap(revs[i]);
					e.fwd.shrink_to_fit(); e.rev.shrink_to_fit();
					stats.cached_overlaps += e.fwd.size();
				}
				e.ready = true;
			}
			divStats.insert(divStats.end(), st.begin(), st.end());
			cvDone.notify_all();
		}

		// ---- quick requests, one device call per (maxOverlaps, forceLocal) class ----
		if (!quickQ.empty())
		{
			std::map<std::pair<int32_t, uint8_t>, std::vector<QuickReq*>> classes;
			size_t taken = 0;
			while (!quickQ.empty() && taken < maxBatch)
			{
				QuickReq* r = quickQ.front(); quickQ.pop_front(); ++taken;
				classes[{r->maxOverlaps, r->forceLocal}].push_back(r);
			}
			const fg_detector_params p = params;
			lk.unlock();
			std::vector<float> allSt;
			size_t calls = 0, reads = 0;
			for (auto& kv : classes)
			{
				std::vector<uint32_t> qids;
				for (QuickReq* r : kv.second) qids.push_back(r->id);
				std::vector<RecList> lists; std::vector<float> st;
				const int rc = deviceCall(p, qids, kv.first.first, kv.first.second, lists, st);
				++calls; reads += qids.size();
				for (size_t i = 0; i < kv.second.size(); ++i)
				{
					kv.second[i]->status = rc;
					if (rc == FG_OK) kv.second[i]->out.swap(lists[i]);
				}
				allSt.insert(allSt.end(), st.begin(), st.end());
			}
			lk.lock();
			stats.device_calls += calls; stats.reads_computed += reads;
			for (auto& kv : classes) for (QuickReq* r : kv.second) r->done = true;
			divStats.insert(divStats.end(), allSt.begin(), allSt.end());
			cvDone.notify_all();
		}
	}
	// wake whoever still waits
	for (uint32_t id : lazyQ) { Entry& e = *cache[id]; e.status = FG_ERR_STATE; e.ready = true; }
	for (QuickReq* r : quickQ) { r->status = FG_ERR_STATE; r->done = true; }
	lazyQ.clear(); quickQ.clear();
	cvDone.notify_all();
}

extern "C" {

int fgb_create(fgb_container** out, fg_ctx* ctx, const struct fg_detector_params* params,
			   uint32_t max_batch, uint32_t linger_us)
{
	if (!out || !ctx || !params) return FG_ERR_ARG;
	if (params->keep_alignment || params->partition_bad_mappings) return FG_ERR_UNSUPPORTED;
	try
	{
		fgb_container* c = new fgb_container;
		c->ctx = ctx; c->params = *params;
		c->maxBatch = max_batch ? max_batch : 4096;
		c->lingerUs = linger_us;
		c->worker = std::thread([c] { c->run(); });
		*out = c;
		return FG_OK;
	}
	catch (...) { return FG_ERR_NOMEM; }
}

void fgb_destroy(fgb_container* c)
{
	if (!c) return;
	{ std::lock_guard<std::mutex> g(c->mu); c->stop = true; }
	c->cvWork.notify_all();
	if (c->worker.joinable()) c->worker.join();
	delete c;
}

int fgb_lazy(fgb_container* c, uint32_t read_id, const struct fg_overlap_rec** recs, uint64_t* n)
{
	if (!c || !recs || !n) return FG_ERR_ARG;
	const uint32_t fwd = read_id & ~1u;	// forward records have even ids (sequence_container.h:27-33)
	std::unique_lock<std::mutex> lk(c->mu);
	++c->stats.requests;
	if (c->stop) return FG_ERR_STATE;
	auto it = c->cache.find(fwd);
	if (it == c->cache.end())
	{
		it = c->cache.emplace(fwd, std::unique_ptr<Entry>(new Entry)).first;
		c->lazyQ.push_back(fwd);
		c->cvWork.notify_one();
	}
	else if (it->second->ready) ++c->stats.cache_hits;
	Entry* e = it->second.get();
	c->cvDone.wait(lk, [&] { return e->ready; });
	if (e->status != FG_OK) return e->status;
	const RecList& l = (read_id & 1u) ? e->rev : e->fwd;
	*recs = l.data(); *n = l.size();
	return FG_OK;
}

int fgb_quick(fgb_container* c, uint32_t read_id, int32_t max_overlaps, uint8_t force_local,
			  struct fg_overlap_rec* out, uint64_t cap, uint64_t* n)
{
	if (!c || !n || (cap && !out) || max_overlaps < 0) return FG_ERR_ARG;
	QuickReq r{read_id, max_overlaps, (uint8_t)(force_local ? 1 : 0)};
	{
		std::unique_lock<std::mutex> lk(c->mu);
		++c->stats.requests;
		if (c->stop) return FG_ERR_STATE;
		c->quickQ.push_back(&r);
		c->cvWork.notify_one();
		c->cvDone.wait(lk, [&] { return r.done; });
	}
	if (r.status != FG_OK) return r.status;
	*n = r.out.size();
	const uint64_t m = r.out.size() < cap ? r.out.size() : cap;
	if (m) memcpy(out, r.out.data(), m * sizeof(fg_overlap_rec));
	return FG_OK;
}

int fgb_prefetch(fgb_container* c, const uint32_t* read_ids, uint32_t n)
{
	if (!c || (n && !read_ids)) return FG_ERR_ARG;
	std::lock_guard<std::mutex> g(c->mu);
	for (uint32_t i = 0; i < n; ++i)
	{
		const uint32_t fwd = read_ids[i] & ~1u;
		if (c->cache.find(fwd) == c->cache.end()) c->prefetchQ.push_back(fwd);
	}
	c->cvWork.notify_one();
	return FG_OK;
}

int fgb_set_divergence_threshold(fgb_container* c, float max_divergence)
{
	if (!c) return FG_ERR_ARG;
	std::lock_guard<std::mutex> g(c->mu);
	c->params.max_divergence = max_divergence;
	return FG_OK;
}

uint64_t fgb_divergence_stats(fgb_container* c, float* out, uint64_t cap)
{
	if (!c) return 0;
	std::lock_guard<std::mutex> g(c->mu);
	const uint64_t m = c->divStats.size() < cap ? c->divStats.size() : cap;
	if (out && m) memcpy(out, c->divStats.data(), m * sizeof(float));
	return c->divStats.size();
}

void fgb_get_stats(fgb_container* c, struct fgb_stats* out)
{
	if (!c || !out) return;
	std::lock_guard<std::mutex> g(c->mu);
	*out = c->stats;
}

} // extern "C"
