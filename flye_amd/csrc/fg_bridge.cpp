// Host-side batch scheduler above the C ABI (include/flye_gpu_bridge.h):
// many caller threads, one dispatcher thread that owns the fg_ctx.
//
// Mirrors the query side of the reference's OverlapContainer (src/sequence/overlap.cpp:
// 518-574): lazySeqOverlaps = cached forward list + complemented twin, quickSeqOverlaps =
// uncached, with the caller's maxOverlaps / forceLocal; fgb_quick_ex = one getSeqOverlaps call
// (overlap.cpp:99-508) with everything it returns for that read, also for records of a container
// the device does not hold (their sequence travels with the request).
#include "../../include/flye_gpu_bridge.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
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

// everything one getSeqOverlaps call returns for one read
struct ReadResult {
	RecList recs;
	std::vector<uint64_t> matchOff;
	std::vector<int32_t> matches;
	std::vector<uint8_t> trim;
	std::vector<float> stats;
};

struct QuickReq {
	uint32_t id; int32_t maxOverlaps; uint8_t forceLocal;
	const uint64_t* words = nullptr; int32_t len = 0;	// foreign record: its sequence
	bool done = false; int status = FG_OK;
	std::unique_ptr<ReadResult> res;
};

// Read-ahead of the quick path, per (maxOverlaps, forceLocal) class.  The reference's callers walk their
// container in id order from N worker threads (processInParallel hands out consecutive indices:
// OverlapContainer::findAllOverlaps overlap.cpp:586-601, estimateOverlaperParameters :786-806, ReadAligner::
// alignReads read_aligner.cpp:195-217), so with one getSeqOverlaps call per thread in flight a device call would
// carry N reads.  Every device call therefore also computes the `ahead` records that follow the highest id asked
// for (same strand); their results wait here until their own call arrives.  `ahead` doubles while the results
// are being picked up and halves when they are not (a caller that jumps around): the cost of a wrong guess is
// device work nobody reads, never a wrong answer.
//
// The assemble stage asks differently: Extender::extendDisjointig takes the overlaps of the current read and then
// asks for the reads on the other side of them, longest overlap first, until one qualifies (extender.cpp:44-98),
// and moves on to that read.  For the lazy class (maxOverlaps 0, not local) the forward records named by the
// overlaps of a read that was asked for are therefore computed ahead as well ("neighbours"), under a budget of
// their own that adapts the same way.
struct SpecEntry { std::unique_ptr<ReadResult> res; uint8_t origin; };		// origin 0: next in id order, 1: neighbour
struct SpecClass {
	std::unordered_map<uint32_t, SpecEntry> ready;	// by record id
	std::deque<uint32_t> order;			// insertion order, for eviction
	uint32_t frontier[2] = {0, 0};		// highest id computed so far, per strand (0 = none)
	uint32_t ahead = 32, aheadNb = 256;
	uint64_t hitsSince = 0, hitsNbSince = 0;	// results picked up since the last speculative call, by origin
	uint64_t computedNbSince = 0, computedSince = 0;
	std::deque<uint32_t> neighbours;	// forward ids named by results handed out, not computed yet
	std::unordered_map<uint32_t, uint8_t> seen;	// ids ever computed or queued in this class (bounds repeated work)
	bool wanted = false;				// low-water mark reached: compute more without waiting for a miss
	// Bulk mode.  A caller that keeps asking for records of the indexed container in an order nobody can guess
	// (Extender::assembleDisjointigs starts from the reads in HASH order, extender.cpp:376-381, and walks the overlap
	// graph from there; ChimeraDetector::estimateGlobalCoverage samples ~1000 reads with rand(), chimera.cpp:55-104)
	// pays one small device call -- milliseconds -- per request, while ALL forward reads of the container cost the
	// device tens of milliseconds.  After `bulkTrigger` requests in a class the dispatcher therefore computes every
	// forward record of the container for that class, `bulkBatch` per call, and keeps the results until they are
	// asked for (at most `bulkMaxRecs` overlaps waiting).
	uint64_t requests = 0;
	bool bulk = false;
	uint32_t bulkNext = 0;				// next record the bulk pass looks at: the forward records first (0 .. n), then -- if
										// reverse-complement records were asked for in this class -- those (n .. 2n)
	uint64_t rcRequests = 0;
	std::vector<uint8_t> done;			// per record (id - first id): a result of this class exists (stored or handed out)
	uint64_t readyRecs = 0;				// overlaps in `ready`
};

} // namespace

// An fg_ctx serves one host thread at a time (flye_gpu.h).  Several containers may sit on one context -- two
// detectors with different parameters over one VertexIndex -- and each has its own dispatcher thread: their device
// calls (fg_set_queries .. fg_overlaps .. fg_set_queries) take turns through one mutex per context.
static std::shared_ptr<std::mutex> contextMutex(fg_ctx* ctx)
{
	static std::mutex regMu;
	static std::map<fg_ctx*, std::weak_ptr<std::mutex>> reg;
	std::lock_guard<std::mutex> g(regMu);
	for (auto it = reg.begin(); it != reg.end();) { if (it->second.expired()) it = reg.erase(it); else ++it; }
	std::shared_ptr<std::mutex> m = reg[ctx].lock();
	if (!m) { m = std::make_shared<std::mutex>(); reg[ctx] = m; }
	return m;
}

struct fgb_container {
	fg_ctx* ctx = nullptr;
	std::shared_ptr<std::mutex> ctxMu;	// shared by every container on this context
	fg_detector_params params;
	uint32_t maxBatch = 4096, lingerUs = 200;
	uint32_t firstId = 0, nFwd = 0, qFirstId = 0, qNFwd = 0;	// id ranges of the context's containers

	std::mutex mu;
	std::condition_variable cvWork, cvDone;
	std::unordered_map<uint32_t, std::shared_ptr<Entry>> cache;	// by forward id (waiters hold the entry too)
	std::deque<uint32_t> lazyQ;		// forward ids somebody waits for
	std::deque<uint32_t> prefetchQ;	// forward ids nobody waits for yet
	std::deque<QuickReq*> quickQ;
	std::map<std::pair<int32_t, uint8_t>, SpecClass> spec;
	uint32_t maxAhead = 4096;			// FGB_READ_AHEAD (0 switches the read-ahead off)
	uint32_t bulkTrigger = 64, bulkBatch = 4096;	// FGB_BULK_TRIGGER (0 = never), FGB_BULK_BATCH
	uint64_t bulkMaxRecs = 64ULL << 20;				// FGB_BULK_MAX_RECS
	std::vector<float> divStats;
	fgb_stats stats{0, 0, 0, 0, 0, 0, 0};
	bool stop = false;
	std::thread worker;
	double tDevice = 0, tConvert = 0;	// dispatcher: seconds inside fg_overlaps / turning batches into per-read results

	void run();
	int deviceCall(const fg_detector_params& p, const std::vector<uint32_t>& ids, int32_t mo, uint8_t fl,
				   std::vector<std::unique_ptr<ReadResult>>& out);
	// a query id the device call would accept (fg_overlaps rejects the whole batch otherwise)
	bool validQueryId(uint32_t id) const
	{
		const uint32_t base = qNFwd ? qFirstId : firstId, cnt = qNFwd ? qNFwd : nFwd;
		return id >= base && id - base < 2 * cnt;
	}
	bool inIndexed(uint32_t id) const { return id >= firstId && id - firstId < 2 * nFwd; }
	bool specWanted() const
	{
		for (auto& kv : spec) if (kv.second.wanted) return true;
		return false;
	}
	void dropSpeculated() { spec.clear(); }		// results computed under parameters that no longer hold
	// the reads Extender looks at next, given the overlaps of the read just handed out: all the reads they name
	// (it sorts by curRange, extender.cpp:53-55, and stops at the first that qualifies; the read it moves on to
	// shares most of its neighbours with this one), longest overlap first.  Every record is queued once per class
	// (`seen`), so the work this adds is bounded by one more pass over the container however the caller walks.
	void queueNeighbours(SpecClass& sc, const RecList& recs)
	{
		std::vector<std::pair<int32_t, uint32_t>> cand;
		cand.reserve(recs.size());
		for (const auto& o : recs)
		{
			const uint32_t e = o.ext_id & ~1u;
			if (!validQueryId(e) || sc.seen.count(e)) continue;
			sc.seen[e] = 1;
			cand.push_back(std::make_pair(-(o.cur_end - o.cur_begin), e));
		}
		std::sort(cand.begin(), cand.end());
		// the dozen longest overlaps of the newest read first (several callers walk different places at once and
		// a call's budget is finite), the rest behind everything queued so far
		const size_t top = std::min<size_t>(cand.size(), 12);
		for (size_t i = top; i-- > 0;) sc.neighbours.push_front(cand[i].second);
		for (size_t i = top; i < cand.size(); ++i) sc.neighbours.push_back(cand[i].second);
		while (sc.neighbours.size() > 16 * (size_t)maxAhead) sc.neighbours.pop_back();
	}
};

int fgb_container::deviceCall(const fg_detector_params& p, const std::vector<uint32_t>& ids, int32_t mo, uint8_t fl,
							  std::vector<std::unique_ptr<ReadResult>>& out)
{
	fg_overlap_batch b;
	const auto t0 = std::chrono::steady_clock::now();
	const int rc = fg_overlaps(ctx, &p, ids.data(), (uint32_t)ids.size(), mo, fl, &b);
	const auto t1 = std::chrono::steady_clock::now();
	tDevice += std::chrono::duration<double>(t1 - t0).count();
	if (getenv("FGB_TRACE") && atoi(getenv("FGB_TRACE")) > 1)
		fprintf(stderr, "[fgb] call of %zu reads: %.2f ms (device %.2f ms), %llu records\n", ids.size(),
				std::chrono::duration<double, std::milli>(t1 - t0).count(), rc == FG_OK ? b.device_seconds * 1e3 : 0.0,
				rc == FG_OK ? (unsigned long long)b.n_recs : 0ULL);
	if (rc != FG_OK) return rc;
	out.resize(ids.size());
	for (size_t i = 0; i < ids.size(); ++i)
	{
		std::unique_ptr<ReadResult> r(new ReadResult);
		const uint64_t a = b.query_off[i], e = b.query_off[i + 1];
		r->recs.assign(b.recs + a, b.recs + e);
		if (b.match_off)
		{
			r->matchOff.resize(e - a + 1);
			for (uint64_t j = a; j <= e; ++j) r->matchOff[j - a] = b.match_off[j] - b.match_off[a];
			r->matches.assign(b.matches + 2 * b.match_off[a], b.matches + 2 * b.match_off[e]);
		}
		if (b.needs_trim) r->trim.assign(b.needs_trim + a, b.needs_trim + e);
		r->stats.assign(b.div_stats + b.div_stats_off[i], b.div_stats + b.div_stats_off[i + 1]);
		out[i] = std::move(r);
	}
	fg_release_batch(&b);
	tConvert += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
	return FG_OK;
}

void fgb_container::run()
{
	std::unique_lock<std::mutex> lk(mu);
	while (true)
	{
		cvWork.wait(lk, [&] { return stop || !lazyQ.empty() || !quickQ.empty() || !prefetchQ.empty() || specWanted(); });
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
			cache.emplace(id, std::make_shared<Entry>());
			ids.push_back(id);
		}
		if (!ids.empty())
		{
			const fg_detector_params p = params;
			lk.unlock();
			std::vector<std::unique_ptr<ReadResult>> res;
			int rc;
			{ std::lock_guard<std::mutex> ctxTurn(*ctxMu); rc = deviceCall(p, ids, 0, 0, res); }
			std::vector<RecList> revs(rc == FG_OK ? ids.size() : 0);
			std::vector<float> st;
			for (size_t i = 0; i < revs.size(); ++i)
			{
				revs[i].reserve(res[i]->recs.size());
				for (const auto& o : res[i]->recs) revs[i].push_back(complement(o));
				st.insert(st.end(), res[i]->stats.begin(), res[i]->stats.end());
			}
			lk.lock();
			++stats.device_calls; stats.reads_computed += ids.size();
			for (size_t i = 0; i < ids.size(); ++i)
			{
				Entry& e = *cache[ids[i]];
				e.status = rc;
				if (rc == FG_OK)
				{
					e.fwd.swap(res[i]->recs); e.rev.swap(revs[i]);
					e.fwd.shrink_to_fit(); e.rev.shrink_to_fit();
					stats.cached_overlaps += e.fwd.size();
				}
				e.ready = true;		// a failed entry is handed to its waiters, who then drop it (fgb_lazy)
			}
			divStats.insert(divStats.end(), st.begin(), st.end());
			cvDone.notify_all();
		}

		// ---- quick requests: one device call per (maxOverlaps, forceLocal, native / foreign) class ----
		if (!quickQ.empty() || specWanted())
		{
			std::map<std::tuple<int32_t, uint8_t, bool>, std::vector<QuickReq*>> classes;
			size_t taken = 0;
			while (!quickQ.empty() && taken < maxBatch)
			{
				QuickReq* r = quickQ.front(); quickQ.pop_front(); ++taken;
				classes[std::make_tuple(r->maxOverlaps, r->forceLocal, r->words != nullptr)].push_back(r);
			}
			for (auto& kv : spec)
				if (kv.second.wanted) classes[std::make_tuple(kv.first.first, kv.first.second, false)];	// may be a call of its own
			// the records to compute ahead, class by class (native ids only)
			std::map<std::tuple<int32_t, uint8_t, bool>, std::vector<uint32_t>> aheadIds;
			std::map<std::tuple<int32_t, uint8_t, bool>, size_t> nbStart;	// aheadIds[..][nbStart ..) are neighbours
			if (maxAhead)
				for (auto& kv : classes)
				{
					if (std::get<2>(kv.first)) continue;
					SpecClass& sc = spec[std::make_pair(std::get<0>(kv.first), std::get<1>(kv.first))];
					// adapt over windows of a few calls (results are picked up over the calls that follow theirs): the
					// records computed ahead were (not) asked for
					if (sc.computedSince >= 2 * (uint64_t)sc.ahead)
					{
						if (sc.hitsSince * 2 >= sc.computedSince) sc.ahead = std::min(sc.ahead * 2, maxAhead);
						else if (sc.hitsSince * 8 < sc.computedSince) sc.ahead = std::max(sc.ahead / 2, 8u);
						sc.hitsSince = 0; sc.computedSince = 0;
					}
					sc.wanted = false;
					const uint32_t base = qNFwd ? qFirstId : firstId, cnt = qNFwd ? qNFwd : nFwd;
					std::vector<uint32_t>& ids = aheadIds[kv.first];
					if (sc.bulk)
					{
						// every forward record not computed yet, bulkBatch per call, while the store has room
						uint32_t added = 0;
						if (sc.done.size() < 2 * (size_t)cnt) sc.done.resize(2 * (size_t)cnt, 0);
						const uint32_t bulkEnd = sc.rcRequests ? 2 * cnt : cnt;
						while (sc.readyRecs < bulkMaxRecs && added < bulkBatch && sc.bulkNext < bulkEnd)
						{
							const uint32_t idx = sc.bulkNext++;
							const uint32_t id = idx < cnt ? base + 2u * idx : base + 2u * (idx - cnt) + 1u;
							if (sc.done[id - base] || sc.ready.find(id) != sc.ready.end()) continue;
							bool asked = false;
							for (QuickReq* r : kv.second) asked |= r->id == id;
							if (asked) continue;
							ids.push_back(id); ++added;
						}
						nbStart[kv.first] = ids.size();
						if (sc.bulkNext < bulkEnd && sc.readyRecs < bulkMaxRecs) sc.wanted = true;	// the next batch follows at once
						continue;
					}
					for (int strand = 0; strand < 2; ++strand)
					{
						uint32_t top = 0; bool any = false;
						for (QuickReq* r : kv.second)
							if ((int)(r->id & 1u) == strand) { top = any ? std::max(top, r->id) : r->id; any = true; }
						if (!any && !(kv.second.empty() && sc.frontier[strand])) continue;	// nobody walks this strand
						// from the frontier, unless the callers have moved past it or far back behind it (a new walk)
						uint32_t id = sc.frontier[strand];
						if (any && (top > id || id - top > 16u * maxAhead)) id = top;
						uint32_t added = 0;
						while (added < sc.ahead && (uint64_t)id + 2 < (uint64_t)base + 2ULL * cnt)
						{
							id += 2;
							if (sc.ready.find(id) == sc.ready.end()) { ids.push_back(id); ++added; }
						}
						if (added) sc.frontier[strand] = id;
						sc.computedSince += added;
					}
					// neighbours of the reads handed out so far (lazy class only)
					if (std::get<0>(kv.first) == 0 && !std::get<1>(kv.first))
					{
						if (sc.computedNbSince >= 4 * (uint64_t)sc.aheadNb)
						{
							if (sc.hitsNbSince * 8 >= sc.computedNbSince) sc.aheadNb = std::min(sc.aheadNb * 2, maxAhead);
							else if (sc.hitsNbSince * 64 < sc.computedNbSince) sc.aheadNb = std::max(sc.aheadNb / 2, 8u);
							sc.hitsNbSince = 0; sc.computedNbSince = 0;
						}
						nbStart[kv.first] = ids.size();
						uint32_t added = 0;
						while (added < sc.aheadNb && !sc.neighbours.empty())
						{
							const uint32_t id = sc.neighbours.front(); sc.neighbours.pop_front();
							bool asked = false;
							for (QuickReq* r : kv.second) asked |= r->id == id;
							if (asked || sc.ready.find(id) != sc.ready.end()) continue;
							if (std::find(ids.begin(), ids.end(), id) != ids.end()) continue;
							ids.push_back(id); ++added;
						}
						sc.computedNbSince += added;
					}
				}
			const fg_detector_params p = params;
			lk.unlock();
			std::vector<float> allSt;
			size_t calls = 0, reads = 0, specReads = 0;
			std::map<std::tuple<int32_t, uint8_t, bool>, std::vector<std::unique_ptr<ReadResult>>> aheadRes;
			for (auto& kv : classes)
			{
				const bool foreign = std::get<2>(kv.first);
				std::vector<QuickReq*>& reqs = kv.second;
				std::vector<uint32_t> qids;
				int rc = FG_OK;
				std::lock_guard<std::mutex> ctxTurn(*ctxMu);
				if (foreign)
				{
					// the waiting foreign records become a temporary query container; their device ids
					// lie behind the indexed container's and are mapped back afterwards
					std::vector<uint64_t> words, off(1, 0);
					std::vector<int32_t> len;
					for (QuickReq* r : reqs)
					{
						const size_t nw = ((size_t)r->len + 31) / 32;
						words.insert(words.end(), r->words, r->words + nw);
						off.push_back(words.size());
						len.push_back(r->len);
					}
					uint32_t base = firstId + 2 * nFwd;
					if ((uint64_t)base + 2ULL * reqs.size() > 0xFFFFFFFFULL) base = 0;	// room below the indexed ids instead
					if (words.empty()) words.push_back(0);
					rc = fg_set_queries(ctx, (uint32_t)reqs.size(), words.data(), off.data(), len.data(), base);
					for (size_t i = 0; i < reqs.size(); ++i) qids.push_back(base + 2 * (uint32_t)i);
				}
				else
					for (QuickReq* r : reqs) qids.push_back(r->id);
				const std::vector<uint32_t>& extra = aheadIds[kv.first];
				qids.insert(qids.end(), extra.begin(), extra.end());
				if (qids.empty()) continue;
				std::vector<std::unique_ptr<ReadResult>> res;
				if (rc == FG_OK) rc = deviceCall(p, qids, std::get<0>(kv.first), std::get<1>(kv.first), res);
				if (foreign) (void)fg_set_queries(ctx, 0, nullptr, nullptr, nullptr, 0);
				++calls; reads += qids.size(); specReads += extra.size();
				for (size_t i = 0; i < reqs.size(); ++i)
				{
					reqs[i]->status = rc;
					if (rc != FG_OK) continue;
					if (foreign) for (auto& o : res[i]->recs) o.cur_id = reqs[i]->id;
					allSt.insert(allSt.end(), res[i]->stats.begin(), res[i]->stats.end());
					reqs[i]->res = std::move(res[i]);
				}
				if (rc == FG_OK)
				{
					std::vector<std::unique_ptr<ReadResult>>& keep = aheadRes[kv.first];
					for (size_t i = 0; i < extra.size(); ++i) keep.push_back(std::move(res[reqs.size() + i]));
				}
			}
			lk.lock();
			stats.device_calls += calls; stats.reads_computed += reads; stats.reads_ahead += specReads;
			for (auto& kv : classes) for (QuickReq* r : kv.second) r->done = true;
			if (p.max_divergence == params.max_divergence)		// the threshold may have moved while the device worked
				for (auto& kv : aheadRes)
				{
					SpecClass& sc = spec[std::make_pair(std::get<0>(kv.first), std::get<1>(kv.first))];
					const std::vector<uint32_t>& ids = aheadIds[kv.first];
					const size_t nb0 = nbStart.count(kv.first) ? nbStart[kv.first] : ids.size();
					for (size_t i = 0; i < kv.second.size(); ++i)
					{
						sc.readyRecs += kv.second[i]->recs.size();
						if (sc.bulk && ids[i] >= firstId && ids[i] - firstId < sc.done.size()) sc.done[ids[i] - firstId] = 1;
						sc.ready[ids[i]] = SpecEntry{std::move(kv.second[i]), (uint8_t)(i >= nb0 ? 1 : 0)};
						sc.order.push_back(ids[i]);
						sc.seen[ids[i]] = 1;
					}
					while (!sc.bulk && sc.ready.size() > 2 * (size_t)maxAhead && !sc.order.empty())
					{
						auto old = sc.ready.find(sc.order.front());		// oldest first; ids already picked up are no-ops
						if (old != sc.ready.end()) { sc.readyRecs -= old->second.res->recs.size(); sc.ready.erase(old); }
						sc.order.pop_front();
					}
					while (sc.order.size() > 4 * (size_t)maxAhead) sc.order.pop_front();
				}
			// the records on the other side of the overlaps just handed out (lazy class): next call's neighbours
			if (maxAhead)
				for (auto& kv : classes)
				{
					if (std::get<0>(kv.first) != 0 || std::get<1>(kv.first) || std::get<2>(kv.first)) continue;
					SpecClass& sc = spec[std::make_pair(0, (uint8_t)0)];
					for (QuickReq* r : kv.second)
					{
						if (r->status != FG_OK || !r->res) continue;
						sc.seen[r->id] = 1;
						if (!sc.bulk) queueNeighbours(sc, r->res->recs);
					}
					if (sc.seen.size() > 64 * (size_t)maxAhead + 4 * (size_t)(qNFwd ? qNFwd : nFwd)) sc.seen.clear();
				}
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

static int quickCommon(fgb_container* c, QuickReq& r)
{
	std::unique_lock<std::mutex> lk(c->mu);
	++c->stats.requests;
	if (c->stop) return FG_ERR_STATE;
	if (!r.words && c->maxAhead)
	{
		if (c->bulkTrigger && c->inIndexed(r.id) && !c->qNFwd)
		{
			SpecClass& cls = c->spec[std::make_pair(r.maxOverlaps, r.forceLocal)];
			if (r.id & 1u) ++cls.rcRequests;
			if (++cls.requests >= c->bulkTrigger && !cls.bulk) { cls.bulk = true; cls.wanted = true; c->cvWork.notify_one(); }
		}
		auto sit = c->spec.find(std::make_pair(r.maxOverlaps, r.forceLocal));
		if (sit != c->spec.end())
		{
			SpecClass& sc = sit->second;
			auto it = sc.ready.find(r.id);
			if (it != sc.ready.end())
			{
				sc.readyRecs -= it->second.res->recs.size();
				r.res = std::move(it->second.res);
				if (it->second.origin) ++sc.hitsNbSince; else ++sc.hitsSince;
				sc.ready.erase(it);
				++c->stats.ahead_hits;
				c->divStats.insert(c->divStats.end(), r.res->stats.begin(), r.res->stats.end());
				if (r.maxOverlaps == 0 && !r.forceLocal && !sc.bulk) c->queueNeighbours(sc, r.res->recs);	// what the caller asks for next
				// keep the device ahead of the callers: top up when half of the last call's results are gone
				// (bulk mode: when the store has room again)
				if (sc.bulk) { if (!sc.wanted && sc.bulkNext < (sc.rcRequests ? 2 * c->nFwd : c->nFwd) && sc.readyRecs < c->bulkMaxRecs) { sc.wanted = true; c->cvWork.notify_one(); } }
				else
				if (!sc.wanted && sc.ready.size() * 2 < sc.ahead) { sc.wanted = true; c->cvWork.notify_one(); }
				r.done = true; r.status = FG_OK;
				return FG_OK;
			}
		}
	}
	c->quickQ.push_back(&r);
	c->cvWork.notify_one();
	c->cvDone.wait(lk, [&] { return r.done; });
	return r.status;
}

extern "C" {

int fgb_create(fgb_container** out, fg_ctx* ctx, const struct fg_detector_params* params,
			   uint32_t max_batch, uint32_t linger_us)
{
	if (!out || !ctx || !params) return FG_ERR_ARG;
	try
	{
		std::unique_ptr<fgb_container> c(new fgb_container);
		c->ctx = ctx; c->ctxMu = contextMutex(ctx); c->params = *params;
		c->maxBatch = max_batch ? max_batch : 4096;
		c->lingerUs = linger_us;
		if (getenv("FGB_READ_AHEAD")) c->maxAhead = (uint32_t)std::max(0, atoi(getenv("FGB_READ_AHEAD")));
		if (getenv("FGB_BULK_TRIGGER")) c->bulkTrigger = (uint32_t)std::max(0, atoi(getenv("FGB_BULK_TRIGGER")));
		if (getenv("FGB_BULK_BATCH")) c->bulkBatch = (uint32_t)std::max(1, atoi(getenv("FGB_BULK_BATCH")));
		if (getenv("FGB_BULK_MAX_RECS")) c->bulkMaxRecs = strtoull(getenv("FGB_BULK_MAX_RECS"), nullptr, 10);
		const int rc = fg_container_info(ctx, &c->firstId, &c->nFwd, &c->qFirstId, &c->qNFwd);
		if (rc != FG_OK) return rc;
		fgb_container* raw = c.release();
		raw->worker = std::thread([raw] { raw->run(); });
		*out = raw;
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
	if (getenv("FGB_TRACE"))
		fprintf(stderr, "[fgb] %llu device calls, %llu reads (%llu ahead, %llu picked up): %.3f s in fg_overlaps, %.3f s per-read results\n",
				(unsigned long long)c->stats.device_calls, (unsigned long long)c->stats.reads_computed,
				(unsigned long long)c->stats.reads_ahead, (unsigned long long)c->stats.ahead_hits, c->tDevice, c->tConvert);
	delete c;
}

int fgb_lazy(fgb_container* c, uint32_t read_id, const struct fg_overlap_rec** recs, uint64_t* n)
{
	if (!c || !recs || !n) return FG_ERR_ARG;
	if (c->params.keep_alignment || c->params.partition_bad_mappings) return FG_ERR_UNSUPPORTED;
	// a bad id is this caller's error alone: it never reaches a batch (fg_overlaps rejects a batch as a whole)
	if (!c->validQueryId(read_id)) return FG_ERR_ARG;
	const uint32_t fwd = read_id & ~1u;	// forward records have even ids (sequence_container.h:27-33)
	std::unique_lock<std::mutex> lk(c->mu);
	++c->stats.requests;
	if (c->stop) return FG_ERR_STATE;
	auto it = c->cache.find(fwd);
	if (it == c->cache.end())
	{
		it = c->cache.emplace(fwd, std::make_shared<Entry>()).first;
		c->lazyQ.push_back(fwd);
		c->cvWork.notify_one();
	}
	else if (it->second->ready) ++c->stats.cache_hits;
	const std::shared_ptr<Entry> e = it->second;
	c->cvDone.wait(lk, [&] { return e->ready; });
	if (e->status != FG_OK)
	{
		// failures are not cached: the entry leaves the map (its waiters keep it alive), a later request retries
		auto cur = c->cache.find(fwd);
		if (cur != c->cache.end() && cur->second == e) c->cache.erase(cur);
		return e->status;
	}
	const RecList& l = (read_id & 1u) ? e->rev : e->fwd;
	*recs = l.data(); *n = l.size();
	return FG_OK;
}

int fgb_quick(fgb_container* c, uint32_t read_id, int32_t max_overlaps, uint8_t force_local,
			  struct fg_overlap_rec* out, uint64_t cap, uint64_t* n)
{
	if (!c || !n || (cap && !out) || max_overlaps < 0) return FG_ERR_ARG;
	if (c->params.keep_alignment || c->params.partition_bad_mappings) return FG_ERR_UNSUPPORTED;
	if (!c->validQueryId(read_id)) return FG_ERR_ARG;
	QuickReq r; r.id = read_id; r.maxOverlaps = max_overlaps; r.forceLocal = (uint8_t)(force_local ? 1 : 0);
	const int rc = quickCommon(c, r);
	if (rc != FG_OK) return rc;
	*n = r.res->recs.size();
	const uint64_t m = r.res->recs.size() < cap ? r.res->recs.size() : cap;
	if (m) memcpy(out, r.res->recs.data(), m * sizeof(fg_overlap_rec));
	return FG_OK;
}

int fgb_quick_ex(fgb_container* c, uint32_t read_id, const uint64_t* words, int32_t len,
				 int32_t max_overlaps, uint8_t force_local, struct fgb_result* out)
{
	if (!c || !out || max_overlaps < 0 || (words && len < 0)) return FG_ERR_ARG;
	memset(out, 0, sizeof(*out));
	if (c->params.partition_bad_mappings && max_overlaps != 0) return FG_ERR_UNSUPPORTED;
	if (words && c->qNFwd) return FG_ERR_STATE;	// the temporary query container would replace the caller's own
	if (words ? c->inIndexed(read_id) : !c->validQueryId(read_id)) return FG_ERR_ARG;
	QuickReq r; r.id = read_id; r.maxOverlaps = max_overlaps; r.forceLocal = (uint8_t)(force_local ? 1 : 0);
	r.words = words; r.len = len;
	const int rc = quickCommon(c, r);
	if (rc != FG_OK) return rc;
	ReadResult* res = r.res.release();
	out->n = res->recs.size();
	out->recs = res->recs.data();
	if (c->params.keep_alignment) { out->match_off = res->matchOff.data(); out->matches = res->matches.data(); }
	if (c->params.partition_bad_mappings) out->needs_trim = res->trim.data();
	out->n_div_stats = res->stats.size();
	out->div_stats = res->stats.data();
	out->owner_ = res;
	return FG_OK;
}

void fgb_release_result(struct fgb_result* r)
{
	if (!r) return;
	delete (ReadResult*)r->owner_;
	memset(r, 0, sizeof(*r));
}

int fgb_prefetch(fgb_container* c, const uint32_t* read_ids, uint32_t n)
{
	if (!c || (n && !read_ids)) return FG_ERR_ARG;
	if (c->params.keep_alignment || c->params.partition_bad_mappings) return FG_ERR_UNSUPPORTED;
	for (uint32_t i = 0; i < n; ++i)
		if (!c->validQueryId(read_ids[i])) return FG_ERR_ARG;
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
	if (c->params.max_divergence != max_divergence) c->dropSpeculated();
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

// ---- on-disk text forms -----------------------------------------------------------------------
int64_t fg_overlap_dump(const struct fg_overlap_rec* r, const char* cur_name, const char* ext_name, char* buf, uint64_t cap)
{
	if (!r || !cur_name || !ext_name || (cap && !buf)) return FG_ERR_ARG;
	// operator<<(float) = %g with precision 6 on the value converted to double (overlap.h:227-236)
	const int n = snprintf(buf, (size_t)cap, "%s %d %d %d %s %d %d %d -1 -1 %d %g", cur_name, r->cur_begin, r->cur_end,
						   r->cur_len, ext_name, r->ext_begin, r->ext_end, r->ext_len, r->score, (double)r->seq_divergence);
	return n;
}

int fg_overlap_load(const char* line, struct fg_overlap_rec* r, uint32_t* cur_name_off, uint32_t* cur_name_len,
					uint32_t* ext_name_off, uint32_t* ext_name_len)
{
	if (!line || !r || !cur_name_off || !cur_name_len || !ext_name_off || !ext_name_len) return FG_ERR_ARG;
	memset(r, 0, sizeof(*r));
	// operator>> semantics (overlap.h:238-251): whitespace-separated tokens
	const char* p = line;
	auto token = [&](uint32_t& off, uint32_t& len) -> bool
	{
		while (*p == ' ' || *p == '\t') ++p;
		if (!*p || *p == '\n') return false;
		off = (uint32_t)(p - line);
		while (*p && *p != ' ' && *p != '\t' && *p != '\n') ++p;
		len = (uint32_t)(p - line) - off;
		return true;
	};
	auto integer = [&](int32_t& v) -> bool
	{
		uint32_t o, l;
		if (!token(o, l)) return false;
		char* end = nullptr;
		const long x = strtol(line + o, &end, 10);
		if (end != line + o + l) return false;
		v = (int32_t)x;
		return true;
	};
	int32_t ph1, ph2;
	if (!token(*cur_name_off, *cur_name_len) || !integer(r->cur_begin) || !integer(r->cur_end) || !integer(r->cur_len) ||
		!token(*ext_name_off, *ext_name_len) || !integer(r->ext_begin) || !integer(r->ext_end) || !integer(r->ext_len) ||
		!integer(ph1) || !integer(ph2) || !integer(r->score))
		return FG_ERR_ARG;
	uint32_t o, l;
	if (!token(o, l)) return FG_ERR_ARG;
	char* end = nullptr;
	r->seq_divergence = strtof(line + o, &end);		// istream >> float
	if (end != line + o + l) return FG_ERR_ARG;
	return FG_OK;
}

int64_t fg_alignment_dump(int64_t edge_id, const struct fg_overlap_rec* r, const char* read_name, const char* edge_name,
						  char* buf, uint64_t cap)
{
	if (!r || !read_name || !edge_name || (cap && !buf)) return FG_ERR_ARG;
	const int head = snprintf(buf, (size_t)cap, "\tAln\t%lld\t", (long long)edge_id);
	const uint64_t used = (uint64_t)head < cap ? (uint64_t)head : cap;
	const int64_t rest = fg_overlap_dump(r, read_name, edge_name, buf ? buf + used : buf, cap - used);
	return rest < 0 ? rest : head + rest;
}

int64_t fg_fasta_record(const char* name, const uint64_t* words, int32_t len, char* buf, uint64_t cap)
{
	if (!name || len < 0 || (len && !words) || (cap && !buf)) return FG_ERR_ARG;
	const uint64_t nameLen = strlen(name);
	const uint64_t need = 1 + nameLen + 1 + (uint64_t)len + ((uint64_t)len + 79) / 80;
	if (need > cap) return (int64_t)need;
	char* o = buf;
	*o++ = '>'; memcpy(o, name, nameLen); o += nameLen; *o++ = '\n';
	for (int32_t i = 0; i < len; ++i)
	{
		*o++ = "ACGT"[(words[i >> 5] >> ((i & 31) * 2)) & 3];
		if (i % 80 == 79 || i == len - 1) *o++ = '\n';
	}
	return (int64_t)(o - buf);
}

void fgb_get_stats(fgb_container* c, struct fgb_stats* out)
{
	if (!c || !out) return;
	std::lock_guard<std::mutex> g(c->mu);
	*out = c->stats;
}

} // extern "C"
