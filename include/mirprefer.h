/*
 * mirprefer.h -- C-ABI of libmirprefer.so: the MI355X (gfx950) replacement for the subprocess
 * boundary of miR-PREFeR's candidate -> fold -> predict hot path.
 *
 * The reference has no FFI; the boundary it crosses is `samtools depth | awk`, `samtools faidx`,
 * `samtools view` and `RNALfold -L` subprocesses.  Each entry point below
 * names the reference interface (file:line in /root/reference/miR_PREFeR.py, "MP") it replaces.
 *
 * Conventions: plain C types only; inputs are caller-owned and borrowed for the call; outputs are
 * library-owned host buffers released with mirp_free(); every function returns 0 on success and a
 * negative code on error, with text available from mirp_last_error(); one context per device,
 * calls on a context are not re-entrant, different contexts are independent.
 */
#ifndef MIRPREFER_H
#define MIRPREFER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mirp_ctx mirp_ctx;

/* One structure line of `RNALfold -L` output: "<ss> (<energy/100 %6.2f>) <start %4d>". */
typedef struct {
    int32_t start;   /* 1-based start column as RNALfold prints it */
    int32_t len;     /* strlen of the dot-bracket text (incl. dangle dots) */
    int32_t energy;  /* 0.01 kcal/mol */
    int32_t printed; /* 0 if RNALfold's containment rule suppresses the line */
} MirpFoldLine;

/* One ungapped alignment (`<len>M`), 16 bytes.  Arrays of these are sorted stably by (tid, pos) over the
 * sample-ordered concatenation: the order of the reference's combined sorted BAM (MP:667-713, 807-859). */
typedef struct {
    int32_t tid;      /* contig index in @SQ order (MP:500-509) */
    int32_t pos;      /* 1-based leftmost position */
    uint32_t depth;   /* N of the read id `sample_rA_xN` (MP:242-253) */
    uint16_t len;     /* read length */
    uint8_t strand;   /* 0 '+', 1 '-' (flag & 16, MP:1445) */
    uint8_t sample;   /* index into the sample-name list (MP:3300-3308) */
} MirpAln;

typedef struct { int32_t tid, pos, dp, dm; } MirpDepthPos;        /* one line of bam.depth.cut<CUT> (MP:946-949) */
typedef struct { int32_t tid, start, end, strand; } MirpPeak;      /* [start,end) 1-based; dict_contigs entry (MP:961) */
typedef struct { int32_t start, end, strand, depth; } MirpMature;  /* MP:1484; strand -1 = the (0,0,0,0) fallback (MP:1476) */
/* One FASTA entry handed to the folder (MP:1124-1142, 1184-1192). */
typedef struct {
    int32_t tid, ws, we, strand, loc_s, loc_e;
    int32_t tag;                       /* 0 / 1 = 'L' / 2 = 'R' (MP:1116-1120) */
    int32_t n_peaks; int64_t peak_off; /* header peak list */
    int32_t n_matures; int32_t pad0; int64_t mature_off;
    int64_t seq_off; int32_t seq_len; int32_t pad1;
} MirpWindow;
typedef struct { int32_t tid, start, end, n_windows; int32_t w[2][2]; int64_t peak_first; int32_t n_peaks; int32_t pad; } MirpLocus; /* dict_loci entry (MP:1312-1316) */

#define MIRP_MAX_SAMPLES 255     /* ALIGNMENT_FILEs of one run (MP:3300-3308 has no limit; a record's sample index is 8 bits wide) */
#define MIRP_MAX_MIRNA_PER_WINDOW 8
typedef struct { int32_t n_samples, min_mature_len, max_mature_len, allow_3nt, allow_no_star, minlen; } MirpPredictParams;
/* One entry of check_loci's `miRNAs` list (MP:2296-2343); the structure text is line `line` of the window's fold
 * output, characters [ss_off, ss_off+ss_len). */
typedef struct {
    int32_t window, tid, fold_s, fold_e, mat_s, mat_e, star_s, star_e, strand, has_star;
    int32_t line, ss_off, ss_len;
    int32_t reserved; /* imperfect-star flags (MP:2146-2162): bit0 'max_imperfect_star' key present, bits1-2 which+1, bit3 max > 0 */
    int32_t total_depth_mature, total_depth_star;
} MirpMirna;

int mirp_create(int device, mirp_ctx** out);
void mirp_destroy(mirp_ctx* ctx);
const char* mirp_last_error(const mirp_ctx* ctx);
void mirp_free(void* p);
/* ABI version of this header; bumped on any signature change. */
int mirp_abi_version(void);

/*
 * Which RNALfold the fold entry points reproduce.  The reference runs whatever `RNALfold` is on PATH (MP:3053) and bundles two:
 * dependency/Mac/osx-10.9/RNALfold-2.1.2 (Turner-2004 parameters, dangles = 2: the default here, LDS-resident fast path) and
 * dependency/Linux/x64/RNALfold = ViennaRNA 1.8.5 (Turner-1999 parameters, dangles = 1, multi-component structure strings:
 * compatibility mode, one generic kernel).  Both are bit-exact against the respective binary.  Sticky per context.
 */
#define MIRP_FOLD_MODEL_VIENNA_212 0
#define MIRP_FOLD_MODEL_VIENNA_185 1
int mirp_set_fold_model(mirp_ctx* ctx, int32_t model);

/*
 * Replaces: `RNALfold -L <span>` on a FASTA chunk (MP:3047-3119, command MP:3053, consumer MP:1541-1599).
 * seqs: concatenated sequence bytes (any case, T or U); offsets[n_seqs+1]: byte offsets into seqs.
 * Out (library-owned): lines[n_seqs*max_lines], ss[n_seqs*max_lines*ss_stride] (NUL-terminated texts),
 * n_lines[n_seqs], mfe[n_seqs] (0.01 kcal/mol, RNALfold's final " (%6.2f)" line), status[n_seqs]
 * (0 ok, 1 = more than max_lines structures: the first max_lines are returned and n_lines holds the number the sequence needs,
 * <0 = error for that sequence).
 * Any span: up to 300 (and sequences up to 350 nt) on the LDS-resident kernels, beyond that -- PRECURSOR_LEN up to the reference's limit of 3000
 * (MP:167-184) -- on the generic kernels (tables in a global workspace).  Sequences longer than about 4,000 nt (LDS of one workgroup; 5,000 nt: the
 * 32-bit ranking of interior loops) are refused with -5 and a message, not truncated.
 */
int mirp_fold_batch(mirp_ctx* ctx, const char* seqs, const int64_t* offsets, int32_t n_seqs, int32_t span,
                    int32_t max_lines, MirpFoldLine** lines, char** ss, int32_t* ss_stride, int32_t** n_lines,
                    int32_t** mfe, int32_t** status);

/* The same fold, returning the per-sequence summary only (number of structure lines, minimum free energy, status): the structure text stays
 * on the device and no line is copied back.  mirp_last_fold_kernel_ms / mirp_last_fold_fallbacks then cover the whole call. */
int mirp_fold_batch_summary(mirp_ctx* ctx, const char* seqs, const int64_t* offsets, int32_t n_seqs, int32_t span, int32_t max_lines,
                            int32_t** n_lines, int32_t** mfe, int32_t** status);

/*
 * Replaces: the per-window loop of filter_next_loci / check_loci (MP:2350-2432, 2206-2347) with its two
 * `samtools view` subprocesses per window (MP:2002-2031), given the fold output of mirp_fold_batch.
 * Out: mirnas[n_windows*MIRP_MAX_MIRNA_PER_WINDOW] (first n_mirnas[w] valid per window; entry 0 is the one
 * the reference keeps, MP:2494), status[w] (0 ok, >0 capacity flag).
 */
int mirp_predict_batch(mirp_ctx* ctx, const MirpWindow* windows, int32_t n_windows, const MirpMature* matures, int64_t n_matures,
                       const MirpAln* alns, int64_t n_alns, const MirpFoldLine* lines, const char* ss, int32_t ss_stride,
                       int32_t max_lines, const int32_t* n_lines, const MirpPredictParams* params, MirpMirna** mirnas,
                       int32_t** n_mirnas, int32_t** status);

/* The same filter with the -d bookkeeping of check_loci (dict_why_not_miRNA_reasons, MP:2206-2347): besides the miRNA lists, one int32 record per
 * window and per evaluated (mature, structure) pair -- the get_maturestar_info code (MP:1876-1999) and the expression numbers of the pair; layout
 * and flags as documented at mirp_predict_reasons. */
int mirp_predict_batch_reasons(mirp_ctx* ctx, const MirpWindow* windows, int32_t n_windows, const MirpMature* matures, int64_t n_matures,
                               const MirpAln* alns, int64_t n_alns, const MirpFoldLine* lines, const char* ss, int32_t ss_stride,
                               int32_t max_lines, const int32_t* n_lines, const MirpPredictParams* params, MirpMirna** mirnas,
                               int32_t** n_mirnas, int32_t** status, int32_t** reasons, int64_t* n_reasons, int32_t* reasons_stride);

/* ------------------------------------------------------------------------------------------------
 * Device-resident pipeline: inputs are uploaded once, every stage leaves its results in HBM for the
 * next one, and only what a caller asks for is copied back.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t cutoff;         /* READS_DEPTH_CUTOFF (MP:91, awk rule MP:938, weight cap MP:735) */
    int32_t min_peak_len;   /* 19 (MP:3389, MP:956) */
    int32_t max_gap;        /* MAX_GAP (MP:92, MP:1263) */
    int32_t precursor_len;  /* PRECURSOR_LEN (MP:90, MP:1272-1300, RNALfold -L MP:3053) */
} MirpCandidateParams;

/* Genome: contigs in @SQ order, bytes as in the FASTA (case preserved), concatenated. Replaces `samtools faidx` (MP:1100-1105). */
int mirp_load_genome(mirp_ctx* ctx, int32_t n_contigs, const int64_t* contig_len, const uint8_t* seq_concat);
/* Alignments sorted as described at MirpAln. Replaces the prepare-stage BAMs (MP:772-874) as device-resident records. */
int mirp_load_alignments(mirp_ctx* ctx, const MirpAln* alns, int64_t n_alns);
/* Coverage segments of gapped alignments (records of the MirpAln layout, any order): `samtools depth` counts only the M / = / X bases of an
 * alignment (MP:937-941) while the read bookkeeping of the reference works on POS and len(SEQ).  Such an alignment is
 * one MirpAln record (POS, len(SEQ)) plus segments: strand bit 1 set = subtract the interval [pos, pos+len) (takes the record's own interval
 * back out), clear = add it (one per M / = / X block).  Replaces any segments loaded before; mirp_load_alignments clears them. */
int mirp_load_coverage_segments(mirp_ctx* ctx, const MirpAln* segs, int64_t n_segs);
/* Contig sharding across GPUs (one context per shard, whole contigs per shard): the strand vote of the reference double-counts the first
 * position of the first covered run of every contig except the very first run of the depth file (MP:905-906 + 926-929).  A shard whose first
 * covered contig is preceded, in @SQ order, by a covered contig held by ANOTHER shard sets this to 1 so that its first run is double-counted
 * as in the single-file run; default 0 (the context holds the whole genome). */
int mirp_set_contig_shard(mirp_ctx* ctx, int32_t preceded_by_coverage_elsewhere);
/* Window-level re-balancing of a sharded run (the reference fans PIECES of the candidate list out to its worker processes, not contigs:
 * /root/reference/miR_PREFeR.py:1329-1354, 2468-2480).  After mirp_candidate a rank that holds more windows than its share keeps the first
 * n_keep of them -- mirp_fold, mirp_predict, mirp_get_windows, mirp_get_fold and the text writers then see only those -- and ships the rest
 * to under-loaded ranks (host side: mir-prefer_amd/balance.py), which fold and filter them through mirp_fold_batch / mirp_predict_batch.
 * n_keep must not cut an L/R window pair.  The next mirp_candidate restores the full list. */
int mirp_limit_windows(mirp_ctx* ctx, int64_t n_keep);
/* The exclusive prefix sum every compaction of the candidate stage is built on (kept runs, region heads, window slots ...; the Python loops of
 * MP:877-962 and MP:1246-1371 append to lists instead): out[k] = in[0] + .. + in[k-1] for k = 0 .. n, host arrays in and out, computed on the
 * context's device by the kernels the stage uses (one workgroup up to 16,384 elements, a single-pass look-back scan beyond).  For tests. */
int mirp_excl_scan_i32(mirp_ctx* ctx, const int32_t* in, int64_t n, int64_t* out);
/* Replaces gen_contig_typeA + gen_candidate_region_typeA + dump_loci_seqs_and_alignment_multiprocess
 * (MP:877-962, 1246-1371, 1065-1244). contig_order = contig indices in the order sorted(dict_contigs) visits them (MP:1309). */
int mirp_candidate(mirp_ctx* ctx, const MirpCandidateParams* params, const int32_t* contig_order, int64_t* n_peaks, int64_t* n_loci,
                   int64_t* n_windows);
/* Thresholded depth lines of `samtools depth | awk` (MP:937-949), on demand (checkpoint artefact). */
int mirp_get_depth(mirp_ctx* ctx, MirpDepthPos** depth, int64_t* n_depth);
/* dict_contigs (MP:961) in @SQ order. */
int mirp_get_peaks(mirp_ctx* ctx, MirpPeak** peaks, int64_t* n_peaks);
/* dict_loci (MP:1312-1316) + the peaks array (sorted-contig order) its peak_first/n_peaks index. */
int mirp_get_loci(mirp_ctx* ctx, MirpLocus** loci, int64_t* n_loci, MirpPeak** peaks_sorted, int64_t* n_peaks);
/* FASTA entries + alndump payload (MP:1124-1142, 1137): windows, their header peak lists, matures and sequences
 * (seqs[window.seq_off .. +seq_len)). wpeaks/matures are slot arrays indexed by peak_off / mature_off. */
int mirp_get_windows(mirp_ctx* ctx, MirpWindow** windows, int64_t* n_windows, MirpPeak** wpeaks, int64_t* n_wpeaks, MirpMature** matures,
                     int64_t* n_matures, char** seqs, int64_t* n_seq_bytes);
/* Inspection copy of the per-position read table of gen_loci_alignment_info (MP:1395-1468) that the candidate stage keeps in LDS only: for every
 * window and start position ws + x, x in [0, width), of the window's own strand {length of the most abundant read, its depth, total depth}
 * (first-seen maximum, MP:1457), table[(w*width + x)*3 + k]; zeros where no read starts. */
int mirp_get_window_readtable(mirp_ctx* ctx, int32_t** table, int32_t* width, int64_t* n_windows);
/* Replaces run_fold's RNALfold subprocesses (MP:3047-3119) for the resident windows.  max_lines is the structure-line capacity of the main
 * output buffers; RNALfold itself has no limit (MP:3053), so the windows that produce more lines (tandem repeats) are folded again, alone, at the
 * capacity no window can exceed, into side buffers (mirp_get_fold_overflow) that mirp_predict and mirp_write_fold_text read for those windows. */
int mirp_fold(mirp_ctx* ctx, int32_t span, int32_t max_lines);
/* Fold output of the resident windows, same layout as mirp_fold_batch. */
int mirp_get_fold(mirp_ctx* ctx, MirpFoldLine** lines, char** ss, int32_t* ss_stride, int32_t* max_lines, int32_t** n_lines, int32_t** mfe,
                  int32_t** status);
/* The windows of the last mirp_fold that needed more than max_lines structure lines: windows[n] (indices into the window list), their complete
 * output lines[n*max_lines2], ss[n*max_lines2*ss_stride], n_lines[n].  The slots of these windows in mirp_get_fold hold their first max_lines lines only;
 * n_lines / mfe / status of mirp_get_fold(_summary) are the ones of the complete run. */
int mirp_get_fold_overflow(mirp_ctx* ctx, int32_t** windows, int64_t* n, MirpFoldLine** lines, char** ss, int32_t* ss_stride, int32_t* max_lines2,
                           int32_t** n_lines);
/* Per-window summary of the resident fold output (no structure text): n_lines, mfe, status as in mirp_fold_batch. */
int mirp_get_fold_summary(mirp_ctx* ctx, int32_t** n_lines, int32_t** mfe, int32_t** status, int64_t* n_windows);
/* Writes the fold stage artefact `<prefix>_rnalfoldoutput_<i>` in RNALfold's own text format (MP:3085-3098; consumed by MP:1541-1599):
 * for every resident window the `>` header line taken from fasta_path (the candidate stage's FASTA), the printed structure lines
 * "%s (%6.2f) %4d", the upper-cased T->U sequence and " (%6.2f)". */
int mirp_write_fold_text(mirp_ctx* ctx, const char* fasta_path, const char* out_path);
/* The same file, written behind the caller: the call returns once the fold output has been copied off the device; formatting and writing run in
 * worker threads on host copies only, so the next stage (mirp_predict) may start.  mirp_wait_text joins every pending writer of the context and
 * returns the first error (mirp_destroy waits too). */
int mirp_write_fold_text_async(mirp_ctx* ctx, const char* fasta_path, const char* out_path);
int mirp_wait_text(mirp_ctx* ctx);
/* The candidate stage's text artefacts, formatted by native worker threads from the resident state (contig_names: n_names NUL-terminated names
 * back to back, @SQ order): the thresholded depth file `bam.depth.cut<CUT>` -- `chr\tpos\td+\td-` per position with d+ + d- > cutoff
 * (MP:937-949) -- and the folder's input FASTA `<prefix>.rnalfold.in_<i>.fa`: per resident window the header
 * `>chr:ws-we strand locS-locE {0|L|R} s,e,strand;... M:s-e/strand/depth ...` and the sequence (MP:1124-1142, 1162-1178). */
int mirp_write_depth_text(mirp_ctx* ctx, const char* path, const char* contig_names, int32_t n_names);
int mirp_write_window_fasta(mirp_ctx* ctx, const char* path, const char* contig_names, int32_t n_names);
/* Replaces gen_miRNA_loci_nopredict (MP:2435-2502) for the resident windows: per-window check_loci, the 0/(L,R) pairing
 * of filter_next_loci (MP:2373-2432) and the "first mature only" rule (MP:2494).  Out: result[n_result] in window order,
 * ss_text[n_result*ss_stride] NUL-terminated structure strings, n_passed[n_windows] = len(miRNAs) per FASTA entry, status[n_windows] = 0 or the
 * capacity flag of the filter kernel (2: more structure pieces than its table holds, 3: more candidate matures than its table holds): a caller
 * must treat a non-zero status as an error, the window's candidates were truncated. */
int mirp_predict(mirp_ctx* ctx, const MirpPredictParams* params, MirpMirna** result, int64_t* n_result, char** ss_text, int32_t* ss_stride,
                 int32_t** n_passed, int32_t** status, int64_t* n_windows);
/*
 * -d mode (OUTPUT_DETAILS_FOR_DEBUG): why a region is not reported, replaces the dict_why_not_miRNA_reasons bookkeeping of check_loci
 * (MP:2206-2347) that convert_failure_reasons_list / write_dict_reasons (MP:2505-2567) print.  Returns int32 records of `stride` ints:
 * one per window {window, -1, n_structures, any mature in [min,max], n_passed, 0...} and one per evaluated (mature, structure) pair
 * {window, mature index, structure index, line, ss_off, ss_len, get_maturestar_info code (0 ok, 1..12 = MP:1876-1999's failures),
 *  flags, fold_s, fold_e, star_s, star_e, depth on this strand, antisense, mature, isoform, star, imperfect star[3],
 *  mature-star distance, mature depth per sample[n_samples]};  flags: 1 mature/star too close, 2 star but too few reads on the duplex,
 *  4 no star and ALLOW_NO_STAR_EXPRESSION off, 8 too many start positions, 16 mature+iso ratio < 0.8, 32 mature depth <= 100,
 *  64 not expressed in all samples, 128 passed, 256 no read on the precursor.  Record order is unspecified (sort by the first three fields).
 */
int mirp_predict_reasons(mirp_ctx* ctx, const MirpPredictParams* params, int32_t** records, int64_t* n_records, int32_t* stride);
/* Per-stage device time of the last mirp_candidate / mirp_fold / mirp_predict calls, measured with HIP events on the
 * context's stream: ms[0]=coverage scatter+scan, ms[1]=rest of candidate, ms[2]=fold kernel, ms[3]=predict kernel. */
int mirp_last_timings(mirp_ctx* ctx, double ms[4]);
/* Number of windows of the last fold call that the LDS-resident kernel handed to the generic kernel (window longer than 350 nt,
 * or energies outside the fast path's 16-bit ranges); purely informational -- results are identical either way. */
int64_t mirp_last_fold_fallbacks(mirp_ctx* ctx);
/* Number of windows of the last mirp_fold that needed more than max_lines structure lines and were folded again at full capacity. */
int64_t mirp_last_fold_overflow(mirp_ctx* ctx);
/* Which way the last coverage pass (mirp_candidate / mirp_get_depth) took: 1 = the scan built every tile's difference values from the sorted
 * records in LDS (dense inputs without coverage segments), 0 = atomic scatter into the dense difference arrays; informational, the results are
 * identical (the `samtools depth` replacement, /root/reference/miR_PREFeR.py:877-949). */
int mirp_last_coverage_fused(mirp_ctx* ctx);
/* Pins the coverage path of the following mirp_candidate / mirp_get_depth calls: -1 = picked by record density (default), 0 = atomic scatter,
 * 1 = fused scan whenever the input allows it (no coverage segments, no record longer than a scan tile).  For tests and measurements. */
int mirp_set_coverage_path(mirp_ctx* ctx, int32_t mode);
/* Pins how the fill kernel of the following folds relaxes the multiloop splits: 0 = over split candidates (default; a window whose
 * candidate pool overflows is folded by the dense kernel), 1 = the dense loop for every window.  The tables, and therefore every output, are
 * identical either way (RNALfold's DML minimum, /root/reference/miR_PREFeR.py:3053); for tests and measurements. */
int mirp_set_fold_split_path(mirp_ctx* ctx, int32_t mode);
/* Number of windows of the last fold call that the candidate-pool pass handed to the dense fill kernel (pool overflow or no room for a pool). */
int64_t mirp_last_fold_dense(mirp_ctx* ctx);

/* Device time of the kernels of the last mirp_fold, HIP events on the context's stream: ms[0] = fill kernel(s) (fold_lds_kernel: the
 * dynamic program), ms[1] = epilogue kernel(s) (exterior sweep, enumeration, backtracks), summed over the sub-batches of the main pass. */
int mirp_last_fold_kernel_ms(mirp_ctx* ctx, double ms[2]);
/* Measures the two roofs of the fold's fill kernel on this GPU with its own geometry (one 1024-thread workgroup per CU): out[0] ds_read_b32 and
 * out[1] ds_read_u16 wave-instructions per second (conflict-free, reads in flight), out[2] packed 16-bit add+min and out[3] 32-bit shift-add+min
 * VALU wave-instructions per second.  Measurement only; bench.py prices the fill kernel against them. */
int mirp_microbench(mirp_ctx* ctx, double out[4]);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: one process and one context per GPU, whole contigs per rank, no collective on the data path except
 * the ONE exchange step: the gather of the final loci list to rank 0 -- the analogue of the reference's `multiprocessing.Queue.put(list)` per
 * piece that the parent collects (MP:2461-2499) -- and, in the sharded ingest, the routing of parsed records to the rank that owns their contig
 * (mirp_ingest_sams_shard).  Both run over RCCL on the context's stream.  librccl.so.1 is resolved with dlopen at the first mirp_dist_* call: a
 * single-GPU run never loads it, and the process holds one HIP runtime and one RCCL instance (no torch type or torch communicator is involved;
 * the host binding only has to carry the 128-byte id from rank 0 to the other ranks, through a file or any CPU-side store).
 * ---------------------------------------------------------------------------------------------- */
#define MIRP_DIST_ID_BYTES 128
/* ncclGetUniqueId: called by one rank, the bytes are handed to every rank's mirp_dist_init. */
int mirp_dist_unique_id(uint8_t id[MIRP_DIST_ID_BYTES]);
/* ncclCommInitRank on the context's device (collective over all `world` ranks).  world = 1 is valid (every exchange is then a local copy). */
int mirp_dist_init(mirp_ctx* ctx, const uint8_t id[MIRP_DIST_ID_BYTES], int32_t rank, int32_t world);
/* The same rank / world bookkeeping without RCCL, for ranks that SHARE one GPU (RCCL refuses two ranks on one device): every exchange is staged
 * through the host and files in `dir`, a directory all ranks see.  For tests and for debugging a sharded run on a single-GPU machine. */
int mirp_dist_init_local(mirp_ctx* ctx, const char* dir, int32_t rank, int32_t world);
int mirp_dist_finalize(mirp_ctx* ctx);
int mirp_dist_rank(const mirp_ctx* ctx);
int mirp_dist_world(const mirp_ctx* ctx);
/* What RCCL reports about the context's communicator: info = {ncclCommCount, ncclCommUserRank, ncclCommCuDevice}; {-1, -1, -1} without a RCCL
 * communicator (one rank, or mirp_dist_init_local).  (The reference's pool has no analogue; it lets a scaling record prove N ranks on N devices.)
 * Every wait on a peer inside the mirp_dist_* / mirp_gather_* / mirp_exchange_bytes / mirp_ingest_sams_shard calls has a deadline of
 * MIRP_DIST_TIMEOUT_S seconds (environment, default 600): on expiry, or on an asynchronous RCCL error, the communicator is aborted (ncclCommAbort)
 * and the call -- and every later exchange on the context -- returns -7, where the reference's parent waits for ever on a crashed child
 * (result queue, MP:2461-2499). */
int mirp_dist_comm_info(mirp_ctx* ctx, int32_t info[3]);
/* Sum over the ranks of a small int64 vector, in place (n <= 1024); every rank gets the sums.  mirp_dist_barrier = one such reduction. */
int mirp_dist_allreduce_sum(mirp_ctx* ctx, int64_t* v, int32_t n);
int mirp_dist_barrier(mirp_ctx* ctx);
/* Replaces the result queue of gen_miRNA_loci_nopredict (MP:2461-2499): the result of the last mirp_predict of every rank (records + structure
 * text, as mirp_predict returns them), concatenated in rank order on rank dst straight from the device-resident arrays; n_result = 0 elsewhere.
 * tids are genome-wide contig indices on every rank, so the records need no translation. */
int mirp_gather_loci(mirp_ctx* ctx, int32_t dst, MirpMirna** result, int64_t* n_result, char** ss_text, int32_t* ss_stride);
/* The same gather for arbitrary fixed-size host records (per-rank counts may differ); out = NULL and n_out = 0 on ranks other than dst. */
int mirp_gather_records(mirp_ctx* ctx, const void* rec, int64_t n, int32_t rec_bytes, int32_t dst, void** out, int64_t* n_out);
/* All-to-all of host byte blocks on the context's communicator (RCCL grouped send / recv, staged through the device; the local transport for ranks
 * that share a GPU): send holds the blocks for rank 0 .. world-1 back to back, send_cnt[world] their sizes; *recv (mirp_free) receives the blocks
 * of rank 0 .. world-1 back to back, recv_cnt[world] their sizes.  Carries the window payloads of the re-balancing step (see mirp_limit_windows). */
int mirp_exchange_bytes(mirp_ctx* ctx, const void* send, const int64_t* send_cnt, void** recv, int64_t* recv_cnt);

/* ------------------------------------------------------------------------------------------------
 * Host-side native ingest (no device involved).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_contigs; char* contig_names;   /* n_contigs NUL-terminated names back to back, @SQ order of the first file (MP:500-509) */
    int64_t* contig_len;
    int32_t n_samples; char* sample_names;   /* one per SAM file (MP:3300-3308) */
    MirpAln* alns; int64_t n_alns;           /* stably sorted by (tid, pos) over the sample-ordered concatenation */
    MirpAln* segs; int64_t n_segs;           /* coverage segments of the gapped alignments (see mirp_load_coverage_segments), unordered */
} MirpSamData;
/* Replaces sam2bam / samtools cat / sort / expand_bamfile / strand split of prepare_data (MP:656-746, 772-874): parses the
 * sample SAM files (read ids `sample_rA_xN`) with n_threads threads (0 = all cores) and sorts on the host.  A record keeps POS and len(SEQ),
 * which is what the reference's read bookkeeping uses (MP:1439-1457, 2021); an alignment whose CIGAR is not `<len(SEQ)>M` also yields
 * coverage segments (`samtools depth` counts M / = / X bases only).
 * Returns 0 or -1 with a message in errbuf; release with mirp_free_sam_data. */
int mirp_ingest_sams(const char* const* paths, int32_t n_paths, int32_t n_threads, MirpSamData* out, char* errbuf, size_t errbuf_len);
void mirp_free_sam_data(MirpSamData* data);
/* One keep region of the GFF options as the reference's BED file has it (MP:543-652): contig index, 0-based half-open [start, end). */
typedef struct { int32_t tid, start, end; } MirpRegion;
/* The same ingest with the device doing the record work: host threads tokenize, the GPU applies the keep regions like `samtools view -L`
 * (MP:817-859; n_regions = 0: no filter) and sorts stably by (tid, pos) (LSD radix sort, ties keep sample-then-file order).  The sorted
 * records and segments stay resident as the context's alignments (as after mirp_load_alignments + mirp_load_coverage_segments) and are
 * returned in *out.  seconds (optional): {tokenize, upload + filter, sort, download}.  Errors: mirp_last_error. */
int mirp_ingest_sams_gpu(mirp_ctx* ctx, const char* const* paths, int32_t n_paths, int32_t n_threads, const MirpRegion* keep_regions,
                         int64_t n_regions, MirpSamData* out, double seconds[4]);

/* mirp_ingest_sams_gpu in two halves: the host half (header + threaded tokenizer; no context, so it can run while the device is still being opened) and the
 * device half (keep-region filter, stable (tid, pos) radix sort, the records resident as the context's alignments).  mirp_ingest_tokenized_gpu consumes and
 * releases `tokenized`; mirp_free_tokenized releases one that was never handed over.  Same results as mirp_ingest_sams_gpu (MP:772-874). */
int mirp_tokenize_sams(const char* const* paths, int32_t n_paths, int32_t n_threads, void** tokenized, char* errbuf, size_t errbuf_len);
int mirp_ingest_tokenized_gpu(mirp_ctx* ctx, void* tokenized, const MirpRegion* keep_regions, int64_t n_regions, MirpSamData* out, double seconds[4]);
void mirp_free_tokenized(void* tokenized);

/* The same ingest sharded over the ranks of the context's communicator (mirp_dist_init; without one, or with one rank, it equals
 * mirp_ingest_sams_gpu): every rank tokenizes its own byte range of every SAM file, the records are routed to the rank that owns their
 * contig -- owner_of_tid[n_contigs], the same whole-contig partition the later stages use -- with one all-to-all over RCCL, and every rank
 * filters and sorts what it owns.  Each rank's result is the (tid, pos)-stable subsequence of the single-process result for its contigs
 * (the blocks are laid out in file-then-offset order before the sort), so the two ingests are interchangeable.  Replaces the serial
 * prepare_data of the reference (MP:772-874). */
int mirp_ingest_sams_shard(mirp_ctx* ctx, const char* const* paths, int32_t n_paths, int32_t n_threads, const MirpRegion* keep_regions,
                           int64_t n_regions, const int32_t* owner_of_tid, MirpSamData* out, double seconds[4]);

/* FASTA file -> sequences (replaces the genome access of `samtools faidx`, MP:1100-1105; names are the first word of the header line as
 * faidx has them, bytes and case kept).  names: n_contigs NUL-terminated names back to back in file order; len[k] = length of sequence k or
 * -1 when it was not wanted; seq: the wanted sequences back to back (n_bytes).  want / n_want: keep only these names (n_want = 0: all). */
typedef struct { int32_t n_contigs; char* names; int64_t* len; uint8_t* seq; int64_t n_bytes; } MirpFastaData;
int mirp_read_fasta(const char* path, const char* const* want, int32_t n_want, MirpFastaData* out, char* errbuf, size_t errbuf_len);
void mirp_free_fasta_data(MirpFastaData* data);

/* Report side, host only: the bodies of the per-locus read-mapping files of gen_map_result (MP:2907-2959) -- per sample the
 * precursor with `total_mapped_reads=`, the structure, and every read of the locus' strand that lies inside the precursor laid out under it
 * (`m` / `s` padding and the [mature] / [star] marks for the exact mature / star reads) -- from the position-sorted records and the genome
 * instead of one `samtools view` + `samtools faidx` per locus.  loci[n][8] = {tid, fold_s, fold_e, mat_s, mat_e, star_s, star_e, strand};
 * ss = n NUL-terminated structure strings back to back; contig_seq[t] may be NULL for contigs this process does not hold (a locus there is
 * an error, -2); counts[n][n_samples] = reads on the precursor.  Out: text (bodies back to back) and offsets[n+1], released with mirp_free. */
int mirp_report_readmapping(const int32_t* loci, int64_t n_loci, const char* ss, const MirpAln* alns, int64_t n_alns, const uint8_t* const* contig_seq,
                            const int64_t* contig_len, int32_t n_contigs, const char* sample_names, int32_t n_samples, const int64_t* counts,
                            char** text, int64_t** offsets);


/* Report files of the predict stage, formatted and written by native threads, byte for byte what the reference writes:
 * <prefix>_miRNA.gff3 (gen_gff_from_result MP:2619-2641), _miRNA.mature.fa / _miRNA.precursor.fa / _miRNA.precursor.ss
 * (gen_mirna_fasta_ss_from_result MP:2963-3019), _miRNA.detail.csv (gen_csv_table MP:2744-2779), _miRNA.detail.html (gen_html_table_file
 * MP:2793-2904), miRNA.stat.txt (MP:3585-3593).  Host only (no device).  The loci are in their final order (resultlist.sort(), MP:2622):
 * loci[n][10] = {contig index, fold_s, fold_e, mat_s, mat_e, star_s, star_e, strand (0 '+', 1 '-'), star expressed (0 / 1), overhang
 * (0 "2:2", 1 "2:3", 2 "3:3")}; contig_names / ss / precursor / sample_names = NUL-terminated strings back to back (precursor = forward-strand
 * text chrom:fold_s-(fold_e-1), upper case, T -> U); counts[n][n_samples][4] = reads on precursor / mature / star / antisense region;
 * mirbase_form = four strings: what stands before and behind the mature sequence in the miRBase search form for taxon "Viridiplantae", then for
 * "ALL" (gen_search_miRBase_str MP:2773-2791).  A NULL path skips that file.  0 = ok, < 0 with a message in errbuf. */
int mirp_write_reports(int64_t n_loci, const int32_t* loci, const char* contig_names, int32_t n_contigs, const char* ss, const char* precursor,
                       const char* sample_names, int32_t n_samples, const int64_t* counts, const char* mirbase_form, const char* gff_path,
                       const char* mature_fa_path, const char* precursor_fa_path, const char* ss_path, const char* csv_path, const char* html_path,
                       const char* stat_path, char* errbuf, size_t errbuf_len);

/* Many small files at once -- the per-locus read-mapping files <folder>/miRNA-precursor_<k>.map.txt of gen_map_result (MP:2907-2959), one per
 * locus: paths = n NUL-terminated file names back to back, file k holds text[offs[k] .. offs[k+1]).  Creating thousands of files is system-call
 * time spent in the kernel's directory lock: n_threads <= 1 writes them with one native thread (the default of the Python host; 1 / 2 / 4 / 8 threads
 * measured the same on tmpfs and no better on an overlay file system), n_threads = k splits the list over k.  Host only.  0 = ok, -8 with the first
 * failing name in errbuf. */
int mirp_write_files(int64_t n_files, const char* paths, const char* text, const int64_t* offs, int32_t n_threads, char* errbuf, size_t errbuf_len);

/* The tail of run_predict in ONE call (MP:3545-3627; host only): from a result list as mirp_predict / mirp_gather_loci return it (records + structure
 * text rows of ss_stride bytes) to the readmapping/ folder and the seven report files under outdir.  Inside: the more abundant arm becomes the mature
 * (MP:2611-2617), the list is put in the order of resultlist.sort() (MP:2622: [chr, fold_s, fold_e, mat_s, mat_e, star_s, star_e, ss, strand, has_star],
 * element by element, stable), the per-sample read counts of gen_mirna_info (MP:2644-2728) are taken from the (tid, pos)-sorted records, then
 * mirp_report_readmapping + mirp_write_files (on a thread of their own) and mirp_write_reports do the formatting.  contig_seq[t] / contig_len[t] = the
 * bases of contig t on the host (NULL where not held: a locus there is an error).  Optional outputs (NULL to skip): order_out[n] = input index of
 * list position i, sorted_out[n] = the records in list order after the swap, counts_out[n][n_samples][4].  n = 0 writes nothing (the reference prints
 * "0 miRNA identified. No result files generated.", MP:3547-3550).  0 = ok, < 0 with a message in errbuf. */
int mirp_write_result_reports(const MirpMirna* result, int64_t n, const char* ss_text, int32_t ss_stride, const char* contig_names, int32_t n_contigs,
                              const uint8_t* const* contig_seq, const int64_t* contig_len, const MirpAln* alns, int64_t n_alns, const char* sample_names,
                              int32_t n_samples, const char* mirbase_form, const char* outdir, const char* prefix, int32_t* order_out,
                              MirpMirna* sorted_out, int64_t* counts_out, char* errbuf, size_t errbuf_len);

/* Window view: the fold and the filter (mirp_fold, mirp_predict, mirp_predict_reasons) run on windows [first, first + count) of the candidate stage's
 * list until the view is changed; count < 0 restores the whole list.  Window indices in the results are relative to `first`.  The getters and text
 * writers of the candidate stage are not affected by a view and must not be called while one is active.  (Analogue in the reference: the pieces
 * its folder and filter processes work through one after the other, MP:1329-1354, 2468-2480.) */
int mirp_select_windows(mirp_ctx* ctx, int64_t first, int64_t count);

/* The fold, the filter and the report files of a single-process run as ONE pipelined call (replaces run_fold + run_predict of the `pipeline` verb,
 * MP:3441-3627, when no stage artefact is to be kept): the window list is cut into about n_chunks chunks at places where the windows alone decide
 * the list order of the results (no window before the cut reaches the first window behind it, or the contig changes; an (L, R) pair is never split),
 * every chunk is folded and filtered (mirp_select_windows + mirp_fold + mirp_predict), and a host thread turns a chunk's loci into read-mapping files
 * (mirp_write_result_reports' steps) WHILE THE DEVICE FOLDS THE NEXT CHUNK; the seven report files follow at the end.  Arguments as
 * mirp_write_result_reports; max_lines as mirp_fold.  Out: n_loci, the number of chunks used, device_ms = {fold, filter} summed over the chunks.
 * Afterwards the context holds the whole window list again, without a fold (mirp_fold must run before any getter of the fold stage). */
int mirp_fold_predict_report_stream(mirp_ctx* ctx, int32_t span, int32_t max_lines, const MirpPredictParams* pp, int32_t n_chunks, const char* contig_names,
                                    int32_t n_contigs, const uint8_t* const* contig_seq, const int64_t* contig_len, const MirpAln* alns, int64_t n_alns,
                                    const char* sample_names, int32_t n_samples, const char* mirbase_form, const char* outdir, const char* prefix,
                                    int64_t* n_loci, int32_t* n_chunks_used, double device_ms[2]);

#ifdef __cplusplus
}
#endif
#endif
