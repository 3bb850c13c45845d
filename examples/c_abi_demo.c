/* Plain-C consumer of the C ABI (include/flye_gpu.h): packs a few reads, builds the solid
 * k-mer index, asks for the overlaps of every forward read and prints them in Flye's
 * OverlapRange::dump order of fields.  Build:
 *   gcc -std=c99 -pthread -Iinclude examples/c_abi_demo.c -Lflye_amd/lib -lflyegpu -Wl,-rpath,$PWD/flye_amd/lib -o c_abi_demo
 * Then the same reads are asked for one at a time from 8 threads through the batch scheduler
 * of include/flye_gpu_bridge.h.
 * Without a GPU it prints the error of fg_create and exits with status 2. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "flye_gpu.h"
#include "flye_gpu_bridge.h"

/* worker of the second part: asks the batch scheduler for one read at a time, as a Flye
 * worker thread would (OverlapContainer::lazySeqOverlaps, overlap.cpp:528-574) */
struct worker_arg { fgb_container* oc; int first, step, n; uint64_t recs; int rc; };
static void* worker(void* vp)
{
	struct worker_arg* a = (struct worker_arg*)vp;
	int i;
	for (i = a->first; i < a->n; i += a->step)
	{
		const struct fg_overlap_rec* r; uint64_t n;
		a->rc = fgb_lazy(a->oc, (uint32_t)i, &r, &n);	/* both strands: ids 0 .. 2*NREADS-1 */
		if (a->rc != FG_OK) return NULL;
		a->recs += n;
	}
	return NULL;
}

static uint64_t rng_state = 12345;
static uint32_t rnd(void)
{
	rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
	return (uint32_t)(rng_state >> 33);
}

int main(void)
{
	enum { GENOME = 30000, NREADS = 60, RLEN = 6000 };
	static uint8_t genome[GENOME];
	static uint64_t words[NREADS * ((RLEN + 31) / 32)];
	uint64_t word_off[NREADS + 1];
	int32_t len[NREADS];
	uint32_t ids[NREADS];
	fg_ctx* ctx = NULL;
	struct fg_index_stats st;
	struct fg_detector_params p;
	struct fg_overlap_batch b;
	int rc, i, j;

	printf("flye_gpu ABI version %d\n", fg_abi_version());
	rc = fg_create(&ctx, 0, 17);
	if (rc != FG_OK) { printf("fg_create: %s\n", fg_strerror(rc)); return 2; }

	for (i = 0; i < GENOME; ++i) genome[i] = (uint8_t)(rnd() & 3);
	memset(words, 0, sizeof(words));
	word_off[0] = 0;
	for (i = 0; i < NREADS; ++i)
	{
		uint32_t start = rnd() % (GENOME - RLEN);
		uint64_t* w = words + word_off[i];
		for (j = 0; j < RLEN; ++j)
		{
			uint8_t base = genome[start + j];
			if (rnd() % 100 < 5) base = (uint8_t)((base + 1 + rnd() % 3) & 3);	/* 5 % substitutions */
			w[j / 32] |= (uint64_t)base << ((j % 32) * 2);
		}
		len[i] = RLEN;
		word_off[i + 1] = word_off[i] + (RLEN + 31) / 32;
		ids[i] = 2u * (uint32_t)i;
	}
	rc = fg_set_reads(ctx, NREADS, words, word_off, len, 0);
	if (rc == FG_OK) rc = fg_build_index_solid(ctx, 2, 0.40f, 100, 100.0f, 1.0f, &st);
	if (rc != FG_OK) { printf("index: %s (%s)\n", fg_strerror(rc), fg_last_error(ctx)); return 1; }
	printf("index: %llu k-mers, %llu entries, repetitive frequency %llu\n", (unsigned long long)st.selected_kmers,
		   (unsigned long long)st.index_entries, (unsigned long long)st.repetitive_frequency);

	memset(&p, 0, sizeof(p));
	p.max_jump = 1500; p.min_overlap = 1000; p.max_overhang = 1500; p.only_max_ext = 1; p.max_divergence = 1.0f;
	rc = fg_overlaps(ctx, &p, ids, NREADS, 0, 0, &b);
	if (rc != FG_OK) { printf("overlaps: %s (%s)\n", fg_strerror(rc), fg_last_error(ctx)); return 1; }
	printf("%llu overlaps for %u reads, %llu seed hits\n", (unsigned long long)b.n_recs, b.n_queries,
		   (unsigned long long)b.seed_hits);
	for (i = 0; i < 5 && (uint64_t)i < b.n_recs; ++i)
	{
		const struct fg_overlap_rec* r = &b.recs[i];
		printf("%u %d %d %d %u %d %d %d %d %g\n", r->cur_id, r->cur_begin, r->cur_end, r->cur_len, r->ext_id,
			   r->ext_begin, r->ext_end, r->ext_len, r->score, r->seq_divergence);
	}
	rc = b.n_recs > 0 ? 0 : 1;
	{
		/* the same lists through the scheduler: 8 threads, one read per call, both strands;
		 * a reverse-complement id gets the complemented list of its forward read */
		enum { NT = 8 };
		const uint64_t want = 2 * b.n_recs;
		fgb_container* oc = NULL;
		struct fgb_stats bs;
		pthread_t th[NT];
		struct worker_arg wa[NT];
		uint64_t got = 0;
		fg_release_batch(&b);
		if (fgb_create(&oc, ctx, &p, 32, 200) != FG_OK) { printf("fgb_create failed\n"); return 1; }
		for (i = 0; i < NT; ++i)
		{
			wa[i].oc = oc; wa[i].first = i; wa[i].step = NT; wa[i].n = 2 * NREADS; wa[i].recs = 0; wa[i].rc = FG_OK;
			pthread_create(&th[i], NULL, worker, &wa[i]);
		}
		for (i = 0; i < NT; ++i) { pthread_join(th[i], NULL); got += wa[i].recs; if (wa[i].rc != FG_OK) rc = 1; }
		fgb_get_stats(oc, &bs);
		printf("scheduler: %llu overlaps over both strands (expected %llu) from %llu requests in %llu device calls\n",
			   (unsigned long long)got, (unsigned long long)want, (unsigned long long)bs.requests,
			   (unsigned long long)bs.device_calls);
		if (got != want || bs.device_calls >= (uint64_t)NREADS) rc = 1;
		fgb_destroy(oc);
	}
	fg_destroy(ctx);
	return rc;
}
