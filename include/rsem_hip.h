/* rsem_hip.h -- C ABI of librsem_hip.so: RSEM's EM / Gibbs hot path on MI355X (gfx950).
 *
 * The reference (deweylab/RSEM) has no FFI: its hot path is reached through two executables,
 * rsem-run-em (EM.cpp) and rsem-run-gibbs (Gibbs.cpp), whose in-process seams are the pthread
 * entry points  E_STEP<>(Params{model,reader,hitv,ncpv,mhp,countv})  (EM.cpp:57-60,176-247) and
 * Gibbs(Params{no,nsamples,fo,engine,pme_c,...})  (Gibbs.cpp:29-37,265-353): a CSR shard in,
 * per-shard count vector / per-chain accumulators out.  Each entry point below replaces one of
 * those seams (cited per function).  Conventions:
 *   - plain C types; host pointers are caller-owned and only read/written during the call;
 *   - an opaque ctx owns all device memory of one GPU; calls on one ctx are serialised by the caller;
 *   - every function returns RSEM_OK (0) or a negative rsem_status; rsem_hip_strerror() explains,
 *     rsem_hip_last_error() gives the HIP detail of the calling thread's last failure;
 *   - no exceptions, no exit() across the boundary (the reference exits on error, my_assert.h:89-96;
 *     the CLI wrappers in rsem_amd/csrc/host keep that behaviour on top of these codes);
 *   - there is NO CPU fallback: without a usable gfx950 device every *_create fails with
 *     RSEM_ERR_NODEVICE.
 */
#ifndef RSEM_HIP_H_
#define RSEM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RSEM_OK = 0,
    RSEM_ERR_INVALID = -1,  /* bad argument (NULL, sid out of range, non-monotone row_ptr, ...) */
    RSEM_ERR_HIP = -2,      /* a HIP runtime call failed; see rsem_hip_last_error() */
    RSEM_ERR_NOMEM = -3,    /* device or host allocation failed */
    RSEM_ERR_NODEVICE = -4, /* no such GPU / not a gfx950-class device */
    RSEM_ERR_STATE = -5     /* call sequence error (e.g. values never set) */
} rsem_status;

const char* rsem_hip_strerror(int status);
const char* rsem_hip_last_error(void);
int rsem_hip_device_count(int* n);
/* Properties of a device the host programs size their decisions by: key = "compute_units" | "clock_khz" | "hbm_bytes". */
int rsem_hip_device_info(int device, const char* key, int64_t* value);
/* Initialise the HIP runtime and the device context (callable from a helper thread while inputs are parsed). */
int rsem_hip_warmup(int device);
/* ... and load the device code of the parts a program is going to use, so that the first real launch does not pay for it
 * (the runtime loads a translation unit's code object at its first launch: tens of milliseconds each).  what = OR of: */
#define RSEM_PRELOAD_EM 1
#define RSEM_PRELOAD_MODEL 2
#define RSEM_PRELOAD_GIBBS 4
#define RSEM_PRELOAD_CI 8
int rsem_hip_preload(int device, int what);
/* ABI version of this header: bumped on any signature change. */
int rsem_hip_abi_version(void);
/* Measurement aid ("also measure a device STREAM-copy and report both"): streams `bytes` of HBM
 * `reps` times with 16-byte-per-lane non-temporal wave loads, eight in flight per lane, and reports the best rate seen, in GB/s
 * (1e9 bytes per second; copy counts bytes read + bytes written).  Nothing of the hot path depends on it. */
int rsem_hip_stream_probe(int device, uint64_t bytes, int reps, double* read_GBps, double* copy_GBps);

/* ------------------------------------------------------------------------------------------
 * Communicators of the multi-GPU paths.  The reference's parallelism is pthreads over one address
 * space: its "collectives" are the serial sums countvs[0][j] += countvs[i][j] (EM.cpp:385-389) and
 * pme_c[j] += params[i].pme_c[j] (Gibbs.cpp:372-388).  With the shards / chains on different GPUs
 * those sums are RCCL collectives enqueued on the ctx stream.  One rank per GPU; the ranks may be
 * threads of one process (the CLI programs) or one process each (bench.py: the id travels through
 * torch.distributed).
 * ------------------------------------------------------------------------------------------ */
typedef struct rsem_comm rsem_comm;
#define RSEM_COMM_ID_BYTES 128
/* ncclGetUniqueId: call on one rank, hand the bytes to all the others. */
int rsem_comm_unique_id(char* id /* RSEM_COMM_ID_BYTES */);
/* ncclCommInitRank on `device`; collective over all `world` ranks (call concurrently). */
int rsem_comm_create(rsem_comm** out, int device, int rank, int world, const char* id);
/* A group of `world` ranks inside one process that may share devices (RCCL refuses two ranks on one GPU): the
 * exchange goes through the ranks' device buffers with host barriers.  For exercising the sharded paths on a
 * single-GPU machine; out[world]. */
int rsem_comm_create_local(rsem_comm** out, int world, const int* devices);
int rsem_comm_rank(const rsem_comm* comm);
int rsem_comm_world(const rsem_comm* comm);
/* In-place sum of n doubles at device pointer d_buf over all ranks, ordered on `stream` (tests / tools). */
int rsem_comm_allreduce_f64(rsem_comm* comm, void* d_buf, uint64_t n, void* stream);
int rsem_comm_destroy(rsem_comm* comm);

/* ------------------------------------------------------------------------------------------
 * EM (rsem-run-em).  Replaces: HitContainer<HitType> (HitContainer.h:12-59, SingleHit.h:8-51),
 * init<> sharding (EM.cpp:97-174), E_STEP<> (EM.cpp:176-247) and the count reduction / M step /
 * convergence test of EM<> (EM.cpp:383-416).
 * ------------------------------------------------------------------------------------------ */
typedef struct rsem_em_ctx rsem_em_ctx;

/* E-step kernel variants (rsem_em_set_option "kernel") */
#define RSEM_EM_KERNEL_AUTO 0
#define RSEM_EM_KERNEL_CSR 1   /* RETIRED in round 6 (was: thread-per-read over the CSR as given); rsem_em_set_option refuses it */
#define RSEM_EM_KERNEL_SELL 2  /* RETIRED in round 6 (was: per-slice segmented shuffle reduction, a cross-check); refused */
#define RSEM_EM_KERNEL_LANE 3  /* sliced layout, lane-private runs, LDS-staged theta + LDS count window (default) */

/* Upload one shard: N1 reads, nnz alignments.  row_ptr[N1+1] (row_ptr[0]==0, row_ptr[N1]==nnz),
 * sid[nnz] in 1..M (strand already stripped: HitType::getSid, SingleHit.h:26),
 * conprb[nnz] / ncp[N1] may be NULL when they will be supplied by rsem_em_set_values or computed
 * on the device by the rsem_model_* entry points.  (HitContainer::read, HitContainer.h:63-79.) */
int rsem_em_create(rsem_em_ctx** out, int device, int32_t M, uint64_t N1, uint64_t nnz,
                   const uint64_t* row_ptr, const int32_t* sid, const double* conprb,
                   const double* ncp);
/* Replace the CSR values (hit.setConPrb / ncpv[i], EM.cpp:210,216) in the caller's (file) order. */
int rsem_em_set_values(rsem_em_ctx* ctx, const double* conprb, const double* ncp);
/* The current values in caller order (conprb[nnz], ncp[N1]) -- what rsem_em_create / rsem_em_set_values put there or the model
 * rounds computed (hit.getConPrb() / ncpvecs of EM.cpp:421-458 when it writes imdName.ofg).  Works after option "release_csr"
 * (the values are read back from the planes first). */
int rsem_em_get_values(rsem_em_ctx* ctx, double* conprb, double* ncp);
/* Options (none of them is part of the reference's surface):
 *   "kernel"            RSEM_EM_KERNEL_AUTO or _LANE (the cross-check variants were retired in round 6);
 *   "split_rows"        1 (default) / 0: reads with transcript ids outside the LDS window of their own gene are laid out as a row of
 *                       their in-window alignments plus far entries handled by two passes around the E-step kernel;
 *   "split_policy"      1 (default): the reads that are MOSTLY outside split; 2: every read with an id outside (its split rows then
 *                       run on a stream of their own beside the compact reads: "split_overlap" 1, the default with policy 2) --
 *                       measured equal or slower on the bench's inputs, kept as an option;
 *   "check_every"       rounds between the host's looks at the loop;
 *   "value_bits"        64 (default): the theta-only E step streams the conprb doubles as given; 32: reads whose
 *                       non-zero conprb values span less than 2^value_range_bits are streamed as 32-bit mantissas with
 *                       one exponent per read (value = m * 2^e, rounded to nearest: relative error <= 2^-33 of the
 *                       read's largest value), all other reads stay doubles.  Affects the counts of rsem_em_step /
 *                       rsem_em_run / rsem_em_expected_weights; the weights w[] always come from the doubles.
 *   "value_range_bits"  0..24, default 8;
 *   "release_csr"       1: free the caller-order transcript ids and values on the device (12 bytes per alignment: half of the
 *                       context's memory) -- the theta-only rounds of rsem_em_step / rsem_em_run stream the sliced layout
 *                       alone -- until something needs them again (weights, new values, a rebuild of the layout, a model
 *                       context), which reads them back from the planes: the same doubles.  RSEM_ERR_STATE, nothing changed,
 *                       where the planes do not hold everything (Q32 planes, split rows, reads with more than 256
 *                       alignments, a live model context).  0: bring them back now. */
int rsem_em_set_option(rsem_em_ctx* ctx, const char* key, int64_t value);
/* Layout facts: "value_bits", "value_range_bits", "reads_q32", "reads_sliced", "reads_long", "value_plane_bytes",
 * "sid_plane_bytes", "slots", "units", "csr_released" (0 / 1), "csr_bytes" (what "release_csr" frees). */
int rsem_em_get_info(const rsem_em_ctx* ctx, const char* key, int64_t* value);
/* Tuning aid (not part of the reference's surface): one E-step launch of the LANE kernel with per-workgroup start/end
 * timestamps (100 MHz clock), out[2u], out[2u+1] in dispatch order; *n_units_io = capacity in, units written out. */
int rsem_em_debug_trace(rsem_em_ctx* ctx, const double* theta, unsigned long long* out, uint32_t* n_units_io);
int rsem_em_destroy(rsem_em_ctx* ctx);

/* One E step + M step.  theta[M+1] in; counts[M+1] out = fractional counts incl. +N0 in bin 0
 * (EM.cpp:385-392), theta_new = counts / sum (EM.cpp:394-398), sum, and the convergence
 * statistics bChange / totNum (EM.cpp:406-413).  Any output pointer may be NULL. */
int rsem_em_step(rsem_em_ctx* ctx, const double* theta, double N0, double* counts,
                 double* theta_new, double* sum, double* bChange, int32_t* totNum);

/* Per-run measurements filled by rsem_em_run when non-NULL (HIP events on the ctx stream). */
typedef struct {
    double total_ms;        /* first E-step launch .. last M-step kernel of the run */
    double estep_ms_sum;    /* sum of per-launch E-step kernel durations */
    int32_t estep_launches; /* number of E-step launches timed */
    int32_t rounds;         /* rounds executed */
    uint64_t algorithmic_bytes_per_round; /* 12*nnz + 16*N1 + 16*(M+1), SURVEY.md section 8(d) */
} rsem_em_profile;

/* Device-resident EM loop for the rounds where the CSR values are frozen (ROUND >= 12 in the
 * reference, EM.cpp:365-416 with needCalcConPrb == updateModel == false): rounds round0+1, ... until
 * (ROUND >= min_round && totNum == 0) || ROUND == max_round.  theta_inout[M+1]; counts[M+1] (last
 * round's counts, may be NULL); rounds_done = last ROUND; bChange/totNum of that round. */
int rsem_em_run(rsem_em_ctx* ctx, double* theta_inout, double N0, int round0, int min_round,
                int max_round, int* rounds_done, double* counts, double* bChange, int32_t* totNum,
                rsem_em_profile* profile);

/* The line the reference prints after every round (EM.cpp:415): ROUND, SUM, bChange, totNum.  When set, rsem_em_run
 * calls fn on the calling thread for every round it executes, in order, shortly after the round finished on the
 * device (the loop itself never waits for the host).  fn = NULL switches it off. */
typedef void (*rsem_em_progress_fn)(int round, double sum, double bChange, int totNum, void* user);
int rsem_em_set_progress(rsem_em_ctx* ctx, rsem_em_progress_fn fn, void* user);

/* Rows sharded over the ranks of `comm` (the reference's thread split, EM.cpp:135-157): every
 * rank holds a ctx over its own reads and the same M; rsem_em_run then must be called on all ranks with the same
 * theta, the GLOBAL N0 and the same round arguments, and sums the fractional counts of every round over the ranks
 * with one all-reduce of M+1 (+ a few) doubles on the ctx stream (EM.cpp:385-389) before the M step, which every
 * rank computes identically.  comm is not owned; NULL detaches. */
int rsem_em_set_comm(rsem_em_ctx* ctx, rsem_comm* comm);
/* The reference's split of the reads over T workers (EM.cpp:135-157): worker i takes rows until it holds >= nHits / T
 * alignments, the last one the rest, every later worker at least one row.  Host-only; bounds[world + 1]. */
int rsem_em_shard_rows(uint64_t N1, const uint64_t* row_ptr, int world, uint64_t* bounds);

/* Final pass with calcExpectedWeights = true (EM.cpp:460-478): counts incl. +N0, posterior weight
 * of every alignment w[nnz] (file order) and of the noise transcript w_noise[N1]; rows whose
 * normaliser is < 1e-300 get zeros (EM.cpp:237-243). */
int rsem_em_expected_weights(rsem_em_ctx* ctx, const double* theta, double N0, double* counts,
                             double* w, double* w_noise);

/* ------------------------------------------------------------------------------------------
 * Read models (rounds 1-11 of rsem-run-em).  Replaces the per-read / per-alignment work of
 * SingleModel / SingleQModel / PairedEndModel / PairedEndQModel: getConPrb + getNoiseConPrb
 * (SingleQModel.h:101-162, PairedEndQModel.h:94-155 and the no-quality twins) as used by
 * E_STEP<> when needCalcConPrb (EM.cpp:210,216) and calcConProbs (EM.cpp:249-278), and
 * update + updateNoise (SingleQModel.h:168-221, PairedEndQModel.h:161-188) when updateModel
 * (EM.cpp:226,233).  The O(table) work between rounds (init / collect / finish / calcMW) stays
 * with the caller (rsem_amd/csrc/host/model_host.hpp).
 * ------------------------------------------------------------------------------------------ */
typedef struct rsem_model_ctx rsem_model_ctx;

/* Immutable inputs, all in the order of imd.dat / imd_alignable*.f[aq] (host pointers, copied). */
typedef struct {
    int32_t model_type;          /* 0 Single, 1 SingleQ, 2 PairedEnd, 3 PairedEndQ */
    int32_t M;
    uint64_t N1, nnz;
    const uint64_t* row_ptr;     /* [N1+1] */
    const int32_t* sid_signed;   /* [nnz] as in .dat: negative = reverse strand (SingleHit.h:24-26) */
    const int32_t* pos;          /* [nnz] */
    const int32_t* insertL;      /* [nnz], paired-end only (PairedEndHit.h:8-34), else NULL */
    /* reads: base ids 0..4 = A C G T N (utils.h:36-50), quality = ASCII - 33 (QProfile.h:44); mate 2 NULL for SE */
    const uint64_t* read_off[2]; /* [N1+1] offsets into read_seq / read_qual */
    const uint8_t* read_seq[2];
    const uint8_t* read_qual[2]; /* NULL for the no-quality models */
    const uint8_t* low_quality;  /* [N1] Read::isLowQuality() after calc_lq (SingleReadQ.h:63-95, PairedEndReadQ.h:55-62) */
    /* references (RefSeq.h): forward-strand base ids incl. poly(A) tail, concatenated */
    const uint64_t* ref_off;     /* [M+2], ref_off[0] = ref_off[1] = 0 */
    const uint8_t* ref_seq;
    const int32_t* fullLen;      /* [M+1] */
    const int32_t* totLen;       /* [M+1] */
    const uint64_t* mask_off;    /* [M+2] offsets (in 32-bit words) into mask_words */
    const uint32_t* mask_words;  /* RefSeq::fmasks */
} rsem_model_data;

/* The current model parameters (one EM round's view of the master model). */
typedef struct {
    double probF;                /* Orientation */
    int32_t seedLen;
    int32_t estRSPD, B;          /* RSPD: pdf/cdf [B+2] */
    const double* rspd_pdf;
    const double* rspd_cdf;
    int32_t gld_lb, gld_ub;      /* LenDist gld: pdf/cdf [ub-lb+1] */
    const double* gld_pdf;
    const double* gld_cdf;
    int32_t has_mld, mld_lb, mld_ub;
    const double* mld_pdf;
    const double* mld_cdf;
    int32_t prof_rows;           /* 100 (QProfile) or proLen (Profile) */
    const double* prof;          /* [prof_rows*25]  p[q or i][ref][read] */
    const double* noise;         /* [100*5] NoiseQProfile p, or [5] NoiseProfile p */
    const double* mw;            /* [M+1] */
} rsem_model_tables;

/* Sums accumulated by one rsem_model_estep_update (the merged helper models of EM.cpp:400-404). */
typedef struct {
    double* prof;                /* [prof_rows*25] */
    double* noise;               /* [100*5] or [5] */
    double* rspd;                /* [B+2] (estRSPD only, else may be NULL) */
    double* gld;                 /* [gld0_ub-gld0_lb+1] over the ORIGINAL support (paired-end only, else NULL) */
    int32_t gld0_lb, gld0_ub;    /* in: support of the helper models' gld (mparams minL-1, maxL) */
} rsem_model_accum;

/* `em` must have been created with the same CSR (|sid_signed|); the model ctx writes the CSR values of
 * `em` on the device and drives its E step. */
int rsem_model_create(rsem_model_ctx** out, rsem_em_ctx* em, const rsem_model_data* data);
int rsem_model_set_tables(rsem_model_ctx* ctx, const rsem_model_tables* t);
/* conprb of every alignment and the noise conprb of every read with the current tables -> em values */
int rsem_model_calc_conprb(rsem_model_ctx* ctx);
/* One round with updateModel = true (EM.cpp:199-236): E step, counts/theta_new/statistics as
 * rsem_em_step, plus the model sufficient statistics weighted by the posterior fractions. */
int rsem_model_estep_update(rsem_model_ctx* ctx, const double* theta, double N0, double* counts,
                            double* theta_new, double* sum, double* bChange, int32_t* totNum,
                            rsem_model_accum* acc);
/* One whole model round (EM.cpp:199-236 with needCalcConPrb = true; updateModel = (acc != NULL)) in ONE pass over the
 * reads: P(read, alignment | transcript) from the tables last set (SingleQModel.h:101-162, PairedEndQModel.h:94-155 and
 * twins), the posterior weights for `theta` (EM.cpp:227,234) and, when `acc` is given, the model's sufficient statistics
 * (SingleQModel.h:168-221, PairedEndQModel.h:161-188); outputs as rsem_em_step.  Equivalent to rsem_model_calc_conprb
 * followed by rsem_model_estep_update (acc != NULL) or rsem_em_step (acc == NULL). */
int rsem_model_round(rsem_model_ctx* ctx, const double* theta, double N0, double* counts, double* theta_new,
                     double* sum, double* bChange, int32_t* totNum, rsem_model_accum* acc /* NULL: no statistics */);
/* current CSR values back to the host in file order (for imd.ofg, EM.cpp:421-457) */
int rsem_model_get_values(rsem_model_ctx* ctx, double* conprb, double* ncp);
int rsem_model_destroy(rsem_model_ctx* ctx);

/* ------------------------------------------------------------------------------------------
 * Gibbs (rsem-run-gibbs).  Replaces: Item / s / hits (Gibbs.cpp:39-47,63-64), Gibbs()
 * (Gibbs.cpp:265-353) incl. sample() (sampling.h:50-65), and the per-chain accumulators that
 * release() sums (Gibbs.cpp:355-388).
 * ------------------------------------------------------------------------------------------ */
typedef struct rsem_gibbs_ctx rsem_gibbs_ctx;

#define RSEM_GIBBS_EXACT 0    /* the reference's sequential collapsed chain, MT19937, bit-identical draws */
#define RSEM_GIBBS_PARALLEL 1 /* data-augmentation sampler (theta | z, then z | theta in parallel), Philox */

/* items CSR incl. the noise column (sid 0), as in .ofg (EM.cpp:435-457).  init_counts[M+1] is 0 or
 * -1 (omitted transcripts, Gibbs.cpp:152-167); alpha[M+1] or NULL (scalar pseudoC); eel/mw[M+1];
 * grp[m+1] gene start indices (GroupInfo.h:34-53). */
int rsem_gibbs_create(rsem_gibbs_ctx** out, int device, int32_t M, uint64_t N1, uint64_t nitems,
                      const uint64_t* row_ptr, const int32_t* sid, const double* conprb,
                      const int32_t* init_counts, const double* alpha, double pseudoC, double totc,
                      uint64_t N0, const double* eel, const double* mw, int32_t m, const int32_t* grp);
/* Per-run measurements filled by rsem_gibbs_run_chains when non-NULL (HIP events on the ctx stream). */
typedef struct {
    double total_ms;   /* first sweep .. last per-sample statistics kernel */
    double sweep_ms;   /* total_ms / sweeps */
    int64_t sweeps;    /* EXACT: rounds (every chain advances in each); PARALLEL: z passes summed over the chains */
    int32_t chains;
    int32_t team;      /* EXACT: workgroups per chain (k_gibbs_exact_team; 1 = one workgroup per chain); 0 otherwise.  (Sits in what
                        * was padding: the layout of the other fields is unchanged.) */
    double reduce_ms;  /* the ONE collective of the multi-GPU split (reduce of the chain sums to rank 0), 0 without a communicator */
} rsem_gibbs_profile;

/* Run nchains independent chains on this GPU -- the reference's worker threads (Gibbs.cpp:207-254): chain k starts
 * from MT19937(seeds[k]) (EXACT) / Philox keyed by seeds[k] (PARALLEL), keeps nsamples[k] samples after `burnin`
 * rounds, one every `gap` rounds.  EXACT: a team of workgroups per chain, all chains in every launch.  PARALLEL: one chain after
 * the other (a sweep fills the GPU).  count_vectors: NULL, or nchains host pointers (each NULL or nsamples[k] x (M+1)
 * int32 = the lines of imd.countvectors<k>, Gibbs.cpp:257-262).  The accumulators receive the SUMS over the kept
 * samples of all chains, added in chain order (release(), Gibbs.cpp:372-388; not yet divided): pme_c, pve_c (sum of
 * squares), pme_tpm, pme_fpkm [M+1], pve_c_genes [m], pve_c_trans [m_trans] (NULL unless allele groups are set).
 * With rsem_gibbs_set_comm the sums are additionally reduced over the communicator and valid on rank 0 only. */
int rsem_gibbs_run_chains(rsem_gibbs_ctx* ctx, int mode, int nchains, const uint32_t* seeds, int burnin,
                          const int32_t* nsamples, int gap, int thin /* PARALLEL only: z passes per round, >= 1 */,
                          int32_t* const* count_vectors, double* pme_c, double* pve_c, double* pme_tpm,
                          double* pme_fpkm, double* pve_c_genes, double* pve_c_trans, rsem_gibbs_profile* prof);
/* One chain: rsem_gibbs_run_chains with nchains = 1 (sweep_ms may be NULL). */
int rsem_gibbs_run(rsem_gibbs_ctx* ctx, int mode, uint32_t seed, int burnin, int nsamples, int gap,
                   int thin, int32_t* count_vectors, double* pme_c, double* pve_c, double* pme_tpm,
                   double* pme_fpkm, double* pve_c_genes, double* sweep_ms /* may be NULL */);
/* Allele-specific references (ref.ta, GroupInfo.h): ta[m_trans+1] = first allele of every transcript.  After this
 * call every rsem_gibbs_run also accumulates, per kept sample, the squared per-transcript count sums
 * (pve_c_trans, Gibbs.cpp:339-345); rsem_gibbs_get_pve_c_trans returns the last run's sums [m_trans]. */
int rsem_gibbs_set_allele_groups(rsem_gibbs_ctx* ctx, int32_t m_trans, const int32_t* ta);
int rsem_gibbs_get_pve_c_trans(rsem_gibbs_ctx* ctx, double* pve_c_trans);
/* Chains sharded over GPUs: every later rsem_gibbs_run_chains ends with ONE reduce of its
 * accumulator sums to rank 0 of `comm` (RCCL over xGMI), replacing the host loop of release() (Gibbs.cpp:372-388).
 * Count vectors need no communication: a rank writes the files of its own chains.  comm is not owned; NULL detaches. */
int rsem_gibbs_set_comm(rsem_gibbs_ctx* ctx, rsem_comm* comm);
int rsem_gibbs_destroy(rsem_gibbs_ctx* ctx);
/* sampling.h:19-44: seeds of the first nchains chains for --seed seed. */
int rsem_gibbs_chain_seeds(uint32_t seed, int nchains, uint32_t* out);

/* ---- credibility intervals (rsem-calculate-credibility-intervals) -----------------------------------------
 * Replaces sample_theta_from_c + Buffer (calcCI.cpp:93-164, Buffer.h:13-80) and calcCI / calcCI_batch
 * (calcCI.cpp:216-388): for every Gibbs count vector, nSpC Dirichlet draws theta ~ Dir(c + pseudoC) / mw,
 * TPM samples (float) and the mean effective length l_bar per draw; then per transcript / gene the shortest
 * interval holding `confidence` of the samples and the coefficient of quartile variation, for TPM and FPKM.
 * The gamma variates come from a counter-based generator (Philox4x32-10 keyed by `seed`), not from the
 * reference's per-thread MT19937 streams: same distribution, different draws (the reference's own output
 * depends on -p).  The interval arithmetic on a given sample row is bit-identical to the reference's. */
typedef struct rsem_ci_profile {
    double sample_ms, sort_ms, interval_ms, total_ms;
    uint64_t n_draws, n_keys_sorted;
} rsem_ci_profile;

/* cvecs: nCV x (M+1) int32 count vectors (imd.countvectors*, Gibbs.cpp:257-262), row-major, index 0 = noise.
 * eel, mw: M+1.  gene_starts: m+1 (ref.grp).  trans_starts: m_trans+1 (ref.ta) or NULL when not allele-specific.
 * Outputs, each 3 x n floats laid out [lb[n] | ub[n] | cqv[n]]: tpm_ci / fpkm_ci n = M (sid 1..M),
 * gene_*_ci n = m, iso_*_ci n = m_trans (NULL when trans_starts is NULL). */
int rsem_ci_calculate(int device, int32_t M, int32_t nCV, int32_t nSpC, const int32_t* cvecs, const double* eel,
                      const double* mw, double pseudoC, uint64_t seed, double confidence, int32_t m,
                      const int32_t* gene_starts, int32_t m_trans, const int32_t* trans_starts, float* tpm_ci,
                      float* fpkm_ci, float* gene_tpm_ci, float* gene_fpkm_ci, float* iso_tpm_ci, float* iso_fpkm_ci,
                      rsem_ci_profile* prof);
/* The same with the samples GIVEN: tpm_samples M x nSamples float (row j-1 = transcript j, TPM), l_bars nSamples -- the layout of
 * the reference's temporary file (Buffer.h:66-80).  For a caller that draws them itself: rsem-calculate-credibility-intervals
 * --ci-stream reference reproduces the reference's per-thread MT19937 + boost gamma draws on the host (csrc/host/ci_stream.hpp:
 * a stream whose consumption depends on the values drawn cannot be cut into device-sized pieces) and hands them over here. */
int rsem_ci_calculate_samples(int device, int32_t M, int32_t nSamples, const float* tpm_samples, const float* l_bars, double confidence,
                              int32_t m, const int32_t* gene_starts, int32_t m_trans, const int32_t* trans_starts, float* tpm_ci,
                              float* fpkm_ci, float* gene_tpm_ci, float* gene_fpkm_ci, float* iso_tpm_ci, float* iso_fpkm_ci,
                              rsem_ci_profile* prof);
/* The sampling stage alone (tests): tpm_samples M x (nCV*nSpC) float, row j-1 = transcript j, as the reference's
 * temporary file (Buffer.h:66-80); l_bars nCV*nSpC. */
int rsem_ci_sample(int device, int32_t M, int32_t nCV, int32_t nSpC, const int32_t* cvecs, const double* eel,
                   const double* mw, double pseudoC, uint64_t seed, float* tpm_samples, float* l_bars);
/* The interval stage alone: nrows rows of nSamples floats (host, not modified) -> lb, ub, cqv [nrows]
 * (calcCI, calcCI.cpp:216-284). */
int rsem_ci_intervals(int device, int64_t nrows, int32_t nSamples, const float* rows, double confidence, float* lb,
                      float* ub, float* cqv);

#ifdef __cplusplus
}
#endif
#endif /* RSEM_HIP_H_ */
