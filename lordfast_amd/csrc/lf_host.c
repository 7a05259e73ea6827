/*
 * lf_host.c -- host side of liblfgpu.so: error state, options, index loading from the reference's
 * on-disk formats, the seed-batch entry point.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "lf_internal.h"

/* ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); kernels of streams that share a queue run
 * one after the other.  The pipeline keeps up to eight chunks in flight, each with its own streams: 16 queues measure
 * best (4 -> 8 queues: +9 % reads/s at four chunks; 24 or more: worse).  Only a default: an explicit setting in the environment wins, and it only takes effect if
 * this library is loaded before the HIP runtime initialises (bench.py sets it first thing as well). */
__attribute__((constructor)) static void lf_default_hw_queues(void) { setenv("GPU_MAX_HW_QUEUES", "16", 0); }


static __thread char g_err[1024];

void lf_set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *lf_last_error(void) { return g_err; }

int lf_device_count(void) { return lfg_device_count(); }

void lf_params_default(lf_params_t *p)
{   /* src/CommandLineParser.cpp:41-55 */
    memset(p, 0, sizeof *p);
    p->min_anchor_len = 14; p->sampling_count = 1000; p->max_map = 10; p->min_read_len = 1000;
    p->max_ref_hits = 1000; p->chain_alg = 0; p->chain_reward = 9.3; p->chain_penalty = 11.4;
    p->gap_penalty = 0.15; p->threads = 0;
}

int lf_params_set_read_group(lf_params_t *P, const char *rg_line)
{   /* src/CommandLineParser.cpp:85-124 */
    if (!P || !rg_line) { lf_set_error("lf_params_set_read_group: bad argument"); return LF_ERR_ARG; }
    if (strstr(rg_line, "@RG") != rg_line) { lf_set_error("SAM read group line does not start with @RG"); return LF_ERR_ARG; }
    if (strstr(rg_line, "\t") != NULL) { lf_set_error("the read group line contained literal <tab> characters. Please replace with escaped tabs: \\t"); return LF_ERR_ARG; }
    if (strlen(rg_line) >= sizeof P->read_group) { lf_set_error("read group line too long"); return LF_ERR_ARG; }
    const char *p; char *q;
    for (p = rg_line, q = P->read_group; *p; p++) {
        if (*p != '\\') *q++ = *p;
        else {
            p++;
            if (*p == 't') *q++ = '\t';
            else if (*p == 'n') *q++ = '\n';
            else if (*p == 'r') *q++ = '\r';
            else if (*p == '\\') *q++ = '\\';
            else if (*p == 0) break;
        }
    }
    *q = 0;
    const char *id = strstr(P->read_group, "ID:");
    if (!id) { lf_set_error("no ID within the read group line"); P->read_group[0] = 0; return LF_ERR_ARG; }
    size_t k = 0;
    for (id += 3; *id && *id != '\n' && *id != '\t' && k + 1 < sizeof P->read_group_id; ) P->read_group_id[k++] = *id++;
    P->read_group_id[k] = 0;
    return LF_OK;
}

void lf_free(void *ptr) { free(ptr); }

static void *read_file(const char *path, size_t *n_out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) { lf_set_error("cannot open %s", path); return NULL; }
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    void *buf = malloc((size_t)sz + 64);
    if (!buf) { fclose(fp); lf_set_error("out of memory reading %s", path); return NULL; }
    size_t got = 0;
    while (got < (size_t)sz) {
        size_t r = fread((char *)buf + got, 1, (size_t)sz - got, fp);
        if (r == 0) break;
        got += r;
    }
    fclose(fp);
    if (got != (size_t)sz) { free(buf); lf_set_error("short read on %s", path); return NULL; }
    memset((char *)buf + sz, 0, 64);
    *n_out = (size_t)sz;
    return buf;
}

/* .ann (lib/bwa/bntseq.c:71-84,108-138): "l_pac n_seqs seed" then per contig "gi name [anno]" / "offset len n_ambs" */
static int load_ann(struct lf_index *ix, const char *path)
{
    FILE *fp = fopen(path, "r");
    if (!fp) { lf_set_error("cannot open %s", path); return LF_ERR_IO; }
    long long lp; int ns; unsigned seed;
    if (fscanf(fp, "%lld%d%u", &lp, &ns, &seed) != 3 || ns <= 0) { fclose(fp); lf_set_error("bad header in %s", path); return LF_ERR_IO; }
    ix->l_pac = lp; ix->n_seqs = ns;
    ix->contigs = (lf_contig_t *)calloc((size_t)ns, sizeof(lf_contig_t));
    for (int i = 0; i < ns; i++) {
        unsigned gi; char name[8192]; int ch, len, nambs; long long offv;
        if (fscanf(fp, "%u%8191s", &gi, name) != 2) goto bad;
        while ((ch = fgetc(fp)) != '\n' && ch != EOF) {}
        if (fscanf(fp, "%lld%d%d", &offv, &len, &nambs) != 3) goto bad;
        ix->contigs[i].offset = offv; ix->contigs[i].len = len; ix->contigs[i].name = strdup(name);
    }
    fclose(fp);
    return LF_OK;
bad:
    fclose(fp);
    lf_set_error("bad record in %s", path);
    return LF_ERR_IO;
}

int lf_index_load(const char *prefix, int device, unsigned flags, lf_index_t **out)
{
    char path[4096];
    size_t n = 0;
    int rc;
    *out = NULL;
    if (lfg_device_count() <= device) { lf_set_error("no gfx950 device %d visible (this library has no CPU path)", device); return LF_ERR_NO_DEVICE; }
    struct lf_index *ix = (struct lf_index *)calloc(1, sizeof *ix);
    ix->device = device; ix->flags = flags;
    uint64_t *bwt_raw = NULL, *sa_raw = NULL, *sa = NULL;

    snprintf(path, sizeof path, "%s.bwt", prefix);          /* lib/bwa/bwt.c:443-462 */
    bwt_raw = (uint64_t *)read_file(path, &n);
    if (!bwt_raw || n < 40) { rc = LF_ERR_IO; goto fail; }
    ix->primary = bwt_raw[0];
    for (int i = 1; i <= 4; i++) ix->L2[i] = bwt_raw[i];
    ix->seq_len = ix->L2[4];
    ix->bwt_size = (n - 40) >> 2;

    snprintf(path, sizeof path, "%s.sa", prefix);           /* lib/bwa/bwt.c:421-441 */
    sa_raw = (uint64_t *)read_file(path, &n);
    if (!sa_raw || n < 56) { rc = LF_ERR_IO; goto fail; }
    if (sa_raw[0] != ix->primary || sa_raw[6] != ix->seq_len) { lf_set_error("SA-BWT inconsistency in %s", path); rc = LF_ERR_IO; goto fail; }
    ix->sa_intv = sa_raw[5];
    if (ix->sa_intv != 32) { lf_set_error("SA interval %llu unsupported (the reference's indexer always writes 32)", (unsigned long long)ix->sa_intv); rc = LF_ERR_IO; goto fail; }
    ix->n_sa = (ix->seq_len + ix->sa_intv) / ix->sa_intv;
    if (n < 56 + (ix->n_sa - 1) * 8) { lf_set_error("%s is truncated", path); rc = LF_ERR_IO; goto fail; }
    sa = (uint64_t *)malloc(ix->n_sa * 8);
    sa[0] = (uint64_t)-1;
    memcpy(sa + 1, sa_raw + 7, (ix->n_sa - 1) * 8);

    snprintf(path, sizeof path, "%s.ann", prefix);
    if ((rc = load_ann(ix, path)) != LF_OK) goto fail;
    if ((uint64_t)ix->l_pac * 2 != ix->seq_len) { lf_set_error("index is not forward+reverse (l_pac %lld, seq_len %llu)", (long long)ix->l_pac, (unsigned long long)ix->seq_len); rc = LF_ERR_IO; goto fail; }

    snprintf(path, sizeof path, "%s.pac", prefix);          /* lib/bwa/bwa.c:270-274 */
    ix->pac = (uint8_t *)read_file(path, &n);
    if (!ix->pac || n < (size_t)(ix->l_pac / 4 + 1)) { rc = LF_ERR_IO; goto fail; }

    rc = lfg_index_upload(ix, (const uint32_t *)(bwt_raw + 5), sa);
    if (rc != LF_OK) goto fail;
    free(bwt_raw); free(sa_raw); free(sa);
    *out = ix;
    return LF_OK;
fail:
    free(bwt_raw); free(sa_raw); free(sa);
    lf_index_free(ix);
    return rc;
}

void lf_index_free(lf_index_t *ix)
{
    if (!ix) return;
    lfg_index_free(ix);
    if (ix->contigs) { for (int i = 0; i < ix->n_seqs; i++) free(ix->contigs[i].name); free(ix->contigs); }
    free(ix->pac);
    free(ix);
}

uint32_t lf_index_genome_len(const lf_index_t *ix) { return (uint32_t)ix->l_pac; }
int lf_index_n_contigs(const lf_index_t *ix) { return ix->n_seqs; }
int lf_index_describe(const lf_index_t *ix, char *buf, size_t cap)
{
    if (!ix || !buf || cap < 2) { lf_set_error("lf_index_describe: bad argument"); return LF_ERR_ARG; }
    return lfg_index_describe(ix, buf, cap);
}
const char *lf_index_contig(const lf_index_t *ix, int i, int64_t *offset, int32_t *len)
{
    if (i < 0 || i >= ix->n_seqs) return NULL;
    if (offset) *offset = ix->contigs[i].offset;
    if (len) *len = ix->contigs[i].len;
    return ix->contigs[i].name;
}

/* ---- stage 1 ---- */
int lf_seed_batch(const lf_index_t *ix, const lf_params_t *p, int n_reads, const char *reads,
                  const uint64_t *off, lf_seeds_t **out)
{
    *out = NULL;
    if (!ix || !p || n_reads < 0 || p->min_anchor_len < 12 || p->min_anchor_len > 20 || p->sampling_count <= 0) {
        lf_set_error("lf_seed_batch: bad argument (k must be in [12,20], c > 0)"); return LF_ERR_ARG;
    }
    if (n_reads == 0) {
        lf_seeds_t *e = (lf_seeds_t *)calloc(1, sizeof *e);
        e->offF = (uint64_t *)calloc(1, 8); e->offR = (uint64_t *)calloc(1, 8);
        e->F = (Seed_t *)malloc(8); e->R = (Seed_t *)malloc(8);
        *out = e;
        return LF_OK;
    }
    lfg_hits_t h;
    int rc = lfg_seed(ix, p, n_reads, reads, off, 1, &h);
    if (rc != LF_OK) return rc;
    lf_seeds_t *s = (lf_seeds_t *)calloc(1, sizeof *s);
    s->n_reads = n_reads;
    s->offF = (uint64_t *)calloc((size_t)n_reads + 1, 8);
    s->offR = (uint64_t *)calloc((size_t)n_reads + 1, 8);
    uint64_t nR = 0;
    for (uint64_t i = 0; i < h.n_hits; i++) nR += h.strand[i];
    s->F = (Seed_t *)malloc((h.n_hits - nR + 1) * sizeof(Seed_t));
    s->R = (Seed_t *)malloc((nR + 1) * sizeof(Seed_t));
    uint64_t f = 0, r = 0;
    for (int i = 0; i < n_reads; i++) {
        s->offF[i] = f; s->offR[i] = r;
        for (uint64_t j = h.read_off[i]; j < h.read_off[i + 1]; j++) {
            Seed_t sd; sd.tPos = h.tpos[j]; sd.qPos = h.qpl[j] & 0xFFFFF; sd.len = h.qpl[j] >> 20;
            if (h.strand[j]) s->R[r++] = sd; else s->F[f++] = sd;
        }
    }
    s->offF[n_reads] = f; s->offR[n_reads] = r;
    s->n_cache = h.counters[0]; s->n_occblk = h.counters[1]; s->n_sa = h.counters[2]; s->n_readbytes = h.counters[3];
    s->ms_search = h.ms_search; s->ms_accept = h.ms_accept; s->ms_locate = h.ms_locate;
    lfg_hits_free(&h);
    *out = s;
    return LF_OK;
}

void lf_seeds_free(lf_seeds_t *s)
{
    if (!s) return;
    free(s->offF); free(s->offR); free(s->F); free(s->R); free(s);
}

/* ---- the ONE place the library reads its environment (INTEGRATION.md lists every name: operational knobs and the hooks the tests use;
 * none of them selects a different algorithm or a CPU path) ---- */
const char *lf_env(const char *name) { return getenv(name); }
long lf_env_long(const char *name, long dflt) { const char *v = getenv(name); return v && *v ? atol(v) : dflt; }
int lf_env_set(const char *name) { return getenv(name) != NULL; }
