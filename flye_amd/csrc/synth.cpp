// Synthetic genome + long-read simulator used by tests/ and bench.py.
//
// Not part of the reference: Flye ships no read simulator and its toy read
// sets are absent from the snapshot.  Everything here is integer
// arithmetic on a counter-free splitmix64 stream, so a (seed, parameters)
// pair regenerates byte-identical reads on any box -- the committed golden
// fixtures under tests/golden/ only hold seeds and expected outputs.
//
// Read layout produced = the layout the C ABI consumes (include/flye_gpu.h):
// forward strand only, 32 nt per uint64 word, nt i at bits (i%32)*2, each
// read starts on a word boundary (same packing as reference
// src/sequence/sequence.h:54-69).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

namespace {

struct Rng {
	uint64_t s;
	explicit Rng(uint64_t seed) : s(seed) {}
	uint64_t next() {
		uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
		return z ^ (z >> 31);
	}
	// uniform in [0, n) -- multiply-high, integer only
	uint64_t below(uint64_t n) {
		return (uint64_t)(((unsigned __int128)next() * n) >> 64);
	}
};

// exp(sigma * z) quantiles (x65536) of a standard normal at (i+0.5)/129
const int32_t Q_S05[129] = {17310, 21073, 23318, 25034, 26467, 27720, 28848, 29884, 30847, 31754, 32613, 33434, 34223, 34983, 35719, 36434, 37130, 37810, 38476, 39129, 39771, 40402, 41025, 41639, 42246, 42846, 43441, 44030, 44615, 45196, 45773, 46347, 46918, 47487, 48054, 48619, 49183, 49746, 50308, 50870, 51432, 51994, 52557, 53120, 53684, 54250, 54817, 55385, 55956, 56528, 57103, 57681, 58261, 58845, 59431, 60022, 60616, 61214, 61817, 62424, 63035, 63652, 64275, 64902, 65536, 66176, 66822, 67475, 68136, 68803, 69479, 70163, 70855, 71557, 72268, 72988, 73719, 74461, 75214, 75979, 76757, 77547, 78352, 79170, 80004, 80854, 81721, 82605, 83508, 84430, 85373, 86338, 87327, 88340, 89379, 90446, 91542, 92670, 93832, 95030, 96267, 97546, 98869, 100241, 101666, 103148, 104692, 106305, 107993, 109764, 111627, 113592, 115673, 117885, 120244, 122774, 125501, 128459, 131693, 135259, 139234, 143724, 148882, 154941, 162278, 171565, 184189, 203813, 248128};
const int32_t Q_S08[129] = {7787, 10668, 12544, 14053, 15362, 16542, 17632, 18655, 19627, 20558, 21456, 22327, 23175, 24004, 24817, 25616, 26404, 27182, 27952, 28715, 29472, 30225, 30973, 31719, 32461, 33203, 33943, 34683, 35423, 36163, 36905, 37648, 38393, 39141, 39891, 40644, 41401, 42162, 42927, 43697, 44472, 45252, 46038, 46830, 47629, 48434, 49246, 50066, 50893, 51729, 52574, 53427, 54290, 55162, 56045, 56939, 57843, 58759, 59687, 60628, 61581, 62548, 63529, 64525, 65536, 66563, 67606, 68666, 69745, 70842, 71958, 73094, 74252, 75432, 76634, 77860, 79112, 80389, 81694, 83028, 84392, 85787, 87215, 88677, 90176, 91714, 93292, 94912, 96577, 98289, 100052, 101868, 103740, 105672, 107668, 109732, 111868, 114082, 116379, 118766, 121249, 123835, 126534, 129356, 132310, 135409, 138667, 142101, 145728, 149570, 153653, 158005, 162662, 167665, 173067, 178929, 185330, 192370, 200176, 208919, 218828, 230227, 243590, 259643, 279593, 305632, 342401, 402608, 551559};
// z quantiles (x65536), same grid: used for the normal (HiFi) length model
const int32_t Q_Z[129] = {-174502, -148715, -135445, -126139, -118844, -112780, -107552, -102930, -98770, -94974, -91472, -88213, -85159, -82280, -79551, -76953, -74471, -72092, -69804, -67598, -65466, -63401, -61397, -59449, -57553, -55703, -53896, -52130, -50401, -48706, -47043, -45410, -43805, -42225, -40670, -39137, -37625, -36133, -34660, -33204, -31764, -30339, -28928, -27531, -26146, -24773, -23410, -22058, -20715, -19380, -18054, -16735, -15423, -14116, -12816, -11520, -10229, -8942, -7658, -6377, -5099, -3823, -2548, -1274, 0, 1274, 2548, 3823, 5099, 6377, 7658, 8942, 10229, 11520, 12816, 14116, 15423, 16735, 18054, 19380, 20715, 22058, 23410, 24773, 26146, 27531, 28928, 30339, 31764, 33204, 34660, 36133, 37625, 39137, 40670, 42225, 43805, 45410, 47043, 48706, 50401, 52130, 53896, 55703, 57553, 59449, 61397, 63401, 65466, 67598, 69804, 72092, 74471, 76953, 79551, 82280, 85159, 88213, 91472, 94974, 98770, 102930, 107552, 112780, 118844, 126139, 135445, 148715, 174502};

int64_t interp(const int32_t* q, Rng& rng)
{
	uint64_t u = rng.below(128ull << 16);
	int i = (int)(u >> 16);
	int64_t f = (int64_t)(u & 0xFFFF);
	return q[i] + (((int64_t)q[i + 1] - q[i]) * f >> 16);
}

struct Sim {
	std::vector<uint8_t> genome;
	std::vector<uint64_t> words;
	std::vector<uint64_t> wordOff;	// n+1
	std::vector<int32_t> len;		// n
	std::vector<int64_t> origin;	// n, template start in genome
	std::vector<uint8_t> strand;	// n
	int64_t totalBases = 0;
};

void mutateCopy(const std::vector<uint8_t>& src, int64_t from, int64_t n,
				std::vector<uint8_t>& dst, int64_t to, int divPermille, Rng& rng)
{
	for (int64_t i = 0; i < n; ++i)
	{
		uint8_t b = src[from + i];
		if (divPermille > 0 && (int)rng.below(1000) < divPermille)
			b = (uint8_t)((b + 1 + rng.below(3)) & 3);
		dst[to + i] = b;
	}
}

} // namespace

extern "C" {

// lenModel: 0 = log-normal sigma 0.5, 1 = log-normal sigma 0.8,
//           2 = normal(mean = medianLen, sd = medianLen * sdPermille / 1000)
struct fs_params {
	uint64_t seed;
	int64_t genomeLen;
	int32_t nRepeatFamilies;	// planted interspersed repeats
	int32_t repeatMinLen, repeatMaxLen;
	int32_t repeatMinCopies, repeatMaxCopies;
	int32_t repeatDivPermille;
	int32_t nHomopolymers;		// runs of 15..60 identical bases
	int32_t nTandems;			// unit 2..7, total 40..400
	int64_t targetBases;		// stop once sum(len) >= targetBases
	int32_t lenModel;
	int32_t medianLen;
	int32_t sdPermille;
	int32_t minLen, maxLen;
	int32_t errPermille10;		// total error rate in 1/10000
	int32_t subPct, insPct;		// del = 100 - sub - ins
	int32_t circular;			// sample reads across the genome end
	uint64_t readSeed;			// != 0: reads are drawn from their own stream (same genome, other reads)
};

void* fs_create(const fs_params* p)
{
	Sim* sim = new Sim;
	Rng rng(p->seed);
	const int64_t G = p->genomeLen;
	sim->genome.resize(G);
	for (int64_t i = 0; i < G; ++i) sim->genome[i] = (uint8_t)(rng.next() >> 62);

	// interspersed repeat families: copy a source segment to random places
	for (int f = 0; f < p->nRepeatFamilies; ++f)
	{
		int64_t L = p->repeatMinLen + (int64_t)rng.below(p->repeatMaxLen - p->repeatMinLen + 1);
		if (L >= G / 4) L = G / 4;
		if (L <= 0) break;
		int copies = p->repeatMinCopies + (int)rng.below(p->repeatMaxCopies - p->repeatMinCopies + 1);
		int64_t src = (int64_t)rng.below(G - L);
		std::vector<uint8_t> unit(sim->genome.begin() + src, sim->genome.begin() + src + L);
		for (int c = 0; c < copies; ++c)
		{
			int64_t to = (int64_t)rng.below(G - L);
			bool rc = rng.below(2);
			if (!rc)
				mutateCopy(unit, 0, L, sim->genome, to, p->repeatDivPermille, rng);
			else
			{
				std::vector<uint8_t> r(L);
				for (int64_t i = 0; i < L; ++i) r[i] = (uint8_t)(3 - unit[L - 1 - i]);
				mutateCopy(r, 0, L, sim->genome, to, p->repeatDivPermille, rng);
			}
		}
	}
	for (int h = 0; h < p->nHomopolymers; ++h)
	{
		int64_t L = 15 + (int64_t)rng.below(46);
		if (L >= G) break;
		int64_t to = (int64_t)rng.below(G - L);
		uint8_t b = (uint8_t)rng.below(4);
		for (int64_t i = 0; i < L; ++i) sim->genome[to + i] = b;
	}
	for (int t = 0; t < p->nTandems; ++t)
	{
		int unit = 2 + (int)rng.below(6);
		int64_t L = 40 + (int64_t)rng.below(361);
		if (L >= G) break;
		int64_t to = (int64_t)rng.below(G - L);
		uint8_t u[8];
		for (int i = 0; i < unit; ++i) u[i] = (uint8_t)rng.below(4);
		for (int64_t i = 0; i < L; ++i) sim->genome[to + i] = u[i % unit];
	}

	// reads
	if (p->readSeed) rng = Rng(p->readSeed);
	sim->wordOff.push_back(0);
	std::vector<uint8_t> buf;
	while (sim->totalBases < p->targetBases)
	{
		int64_t L;
		if (p->lenModel == 2)
			L = p->medianLen + (((int64_t)p->medianLen * p->sdPermille / 1000) * interp(Q_Z, rng) >> 16);
		else
			L = ((int64_t)p->medianLen * interp(p->lenModel == 0 ? Q_S05 : Q_S08, rng)) >> 16;
		L = std::max<int64_t>(p->minLen, std::min<int64_t>(p->maxLen, L));
		if (L > G) L = G;	// a template never wraps more than once
		int64_t start = p->circular ? (int64_t)rng.below(G) : (int64_t)rng.below(G - L + 1);
		bool rc = rng.below(2);
		buf.clear();
		// walk the template, applying errors; L is the template length
		for (int64_t i = 0; i < L; ++i)
		{
			int64_t gp = start + i;
			if (gp >= G) gp -= G;
			uint8_t b = sim->genome[gp];
			if ((int)rng.below(10000) < p->errPermille10)
			{
				int kind = (int)rng.below(100);
				if (kind < p->subPct) buf.push_back((uint8_t)((b + 1 + rng.below(3)) & 3));
				else if (kind < p->subPct + p->insPct)
				{
					buf.push_back((uint8_t)rng.below(4));
					buf.push_back(b);
				}
				// else deletion: emit nothing
			}
			else buf.push_back(b);
		}
		if (buf.empty()) continue;
		if (rc)
		{
			std::reverse(buf.begin(), buf.end());
			for (auto& b : buf) b = (uint8_t)(3 - b);
		}
		size_t n = buf.size();
		size_t nw = (n + 31) / 32;
		size_t w0 = sim->words.size();
		sim->words.resize(w0 + nw, 0);
		for (size_t i = 0; i < n; ++i)
			sim->words[w0 + i / 32] |= (uint64_t)buf[i] << ((i % 32) * 2);
		sim->wordOff.push_back(w0 + nw);
		sim->len.push_back((int32_t)n);
		sim->origin.push_back(start);
		sim->strand.push_back(rc);
		sim->totalBases += (int64_t)n;
	}
	return sim;
}

void fs_destroy(void* h) { delete (Sim*)h; }
int64_t fs_num_reads(void* h) { return (int64_t)((Sim*)h)->len.size(); }
int64_t fs_num_words(void* h) { return (int64_t)((Sim*)h)->words.size(); }
int64_t fs_total_bases(void* h) { return ((Sim*)h)->totalBases; }

void fs_copy(void* h, uint64_t* words, uint64_t* wordOff, int32_t* len,
			 int64_t* origin, uint8_t* strand)
{
	Sim* s = (Sim*)h;
	if (words) memcpy(words, s->words.data(), s->words.size() * 8);
	if (wordOff) memcpy(wordOff, s->wordOff.data(), s->wordOff.size() * 8);
	if (len) memcpy(len, s->len.data(), s->len.size() * 4);
	if (origin) memcpy(origin, s->origin.data(), s->origin.size() * 8);
	if (strand) memcpy(strand, s->strand.data(), s->strand.size());
}

// FASTA writer so the test-side reference dumper can load the same reads
// through the reference's own parser.  Read i is named "r<i>".
int fs_write_fasta(void* h, const char* path, int64_t firstRead, int64_t nReads)
{
	Sim* s = (Sim*)h;
	FILE* f = fopen(path, "w");
	if (!f) return -1;
	int64_t end = (nReads < 0) ? (int64_t)s->len.size()
				: std::min<int64_t>(s->len.size(), firstRead + nReads);
	std::string line;
	for (int64_t r = firstRead; r < end; ++r)
	{
		fprintf(f, ">r%lld\n", (long long)r);
		int32_t n = s->len[r];
		line.resize(n);
		const uint64_t* w = s->words.data() + s->wordOff[r];
		for (int32_t i = 0; i < n; ++i) line[i] = "ACGT"[(w[i / 32] >> ((i % 32) * 2)) & 3];
		fwrite(line.data(), 1, n, f);
		fputc('\n', f);
	}
	fclose(f);
	return 0;
}

} // extern "C"
