// embed_internal.h -- what the translation units of the embedding stage share: the probability source, the tile constants of
// the FAST / symmetric kernels, the loop record, the session object and the launchers of the force kernels.
//   knn_smooth.hip   neighbour selection + neighbour sums from a given matrix (knn_smooth, visualization.py:90-109)
//   embed_fast.hip   FAST forces: row-wise kernel and the symmetric tile kernel (each unordered pair once)
//   embed_seq.hip    SEQ forces: the reference's summation order, bit-pinned (taichi_core.py:305-326)
//   embed.hip        loss reductions, the per-iteration apply kernels, the session API, the drop-in float operators
#pragma once
#include "common.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int EMB_BLK = 256;

// probability source: f32 rows, or u16 sums + LUT (LUT copy in LDS)
struct ProbSrc {
    const float *pf;        // [nrows x ld] or null
    const uint16_t *ps;     // [nrows x ld] or null
    const float *lut;       // device LUT
    int64_t ld;
    int lut_len;
    // SEQ sessions: the sums of local row r are row rowmap[r] of ps (null: row r).  The map never steps back and never skips
    // (rowmap[0] = 0, rowmap[r + 1] - rowmap[r] is 0 or 1): a sample repeats its k-mers in runs, and the rows of a run share ONE
    // stored row -- fewer distinct cache lines under a wave's loads (r05: the force evaluation of C3 1.56 -> 1.3x ms).
    const int32_t *rowmap;
    int64_t src_rows;       // rows of ps (= the session's rows without a map)
};
// stored row of local row r
__device__ __forceinline__ int64_t prob_src_row(const ProbSrc &src, int64_t r) { return src.rowmap ? (int64_t)src.rowmap[r] : r; }

constexpr int F_RPW = 2;          // FAST row-wise kernel: rows per wave
constexpr int F_WAVES = 8;        //   waves per block
constexpr int F_CPL = 8;          // columns per lane per step (16 B of u16 sums / 32 B of f32)
constexpr int F_LUT_LDS = 12416;  // floats of LUT cached in LDS (n_nb^2*k+1 <= 400*31+1 = 12401)

// symmetric FAST kernel: a block = one tile of SY_R rows x SY_C columns, its SY_WAVES waves take SY_RB rows each
constexpr int SY_RB = 64;
constexpr int SY_NRB = 4;
constexpr int SY_R = SY_RB * SY_NRB;     // 256 rows per tile
constexpr int SY_C = KMAP_WAVE * F_CPL;  // 512 columns per tile
constexpr int SY_WAVES = 4;
__host__ __device__ __forceinline__ bool sy_tile_live(int64_t I, int64_t J) {   // some pair of the tile has j > i
    return (J + 1) * SY_C - 1 > I * SY_R;
}

// SEQ kernel geometry
constexpr int SQ_SUB = 4;                       // sub-lanes per row = one DPP quad
constexpr int SQ_ROWS = KMAP_WAVE / SQ_SUB;     // rows per quad wave
constexpr int SQ_WAVES = 4;                     // waves per block (the LUT is staged once per block)

constexpr int MSG_EXTRA = 8;                    // multi-GPU message: six loss limbs, flag, pad behind the 2 N gradient floats
constexpr int MAX_BEST = 64;
struct LoopState {
    long long iters;        // reference iterations executed (loss evaluations)
    int stopped;            // early stop reached (visualization.py:310-311)
    int jitter_used;        // normals consumed from the pre-drawn stream
    int n_best;
    float prev_loss;        // `loss` of the reference loop (inf before the first iteration)
    float last_loss;
    float worst_loss;       // = best_loss[n_best - 1], worst_slot = best_slot[n_best - 1]: with the fields above, all a non-leader
    int worst_slot;         //   thread reads (40 bytes instead of the whole 560-byte record)
    float best_loss[MAX_BEST];   // ascending (bisect.insort_right order)
    int best_slot[MAX_BEST];     // snapshot buffer holding that entry
};

struct kmap_embed {
    int64_t n = 0, row0 = 0, nrows = 0;
    int n_best = 10, mode = KMAP_EMBED_FAST;
    float lr = 0.01f;
    ProbSrc src{};
    float *lut_dev = nullptr;
    float *Y = nullptr, *G = nullptr, *snaps = nullptr, *loss_log = nullptr;
    double *loss_part = nullptr, *loss_sum = nullptr, *normals = nullptr;
    int n_normals = 0, n_part = 0;
    int64_t loss_log_cap = 1 << 16;
    LoopState *states = nullptr;
    int cur = 0;
    bool have_prob = false, have_coords = false;
    // SEQ: rows [0, seq_main_rows) of the local range go to the quad kernel (16 rows per wave), the rest -- the rows that would
    // form a last, nearly empty round of blocks -- to the wide kernel with seq_tail_g lanes per row (0: no tail)
    int64_t seq_main_rows = 0;
    int seq_tail_g = 0;
    int64_t seq_pair_rows = 0;      // of the main rows, [0, seq_pair_rows) run in the pair form (32 rows per wave); multiple of 128
    // SEQ producer / adder form (embed_seq.hip): blocks of seq_R <= seq_RP rows, one adder wave per block (0: the forms above)
    int seq_R = 0, seq_RP = 0;
    // symmetric FAST path (all rows local): partial buffers
    float *rowpart = nullptr, *colpart = nullptr;
    int64_t symI = 0, symJ = 0;
    bool sym = false;
    // symmetric FAST path sharded over ranks: rank r owns the 256-row blocks I = r, r + world, ... (cyclic: the upper-triangle
    // work per block shrinks with I); its probability rows are stored block after block (local block b = I / world)
    int world = 1, rank = 0;
    int64_t n_lblocks = 0;
    // jitter normals: fixed-capacity device buffer + device-resident count, so that the kernel arguments of an iteration never
    // change between launches (a captured hipGraph stays valid when the host refills the pool)
    int *n_normals_dev = nullptr;
    int normals_cap = 0;
    // two iterations (both parities of the double-buffered loop record) captured as one hipGraph and replayed by kmap_embed_step
    hipGraphExec_t gexec = nullptr;
    hipStream_t gstream = nullptr;
    int graph_cur = 0;
    bool graph_failed = false;
};

// ---- launchers (one per force-kernel family); loss partials go to e->loss_part --------------------------------------------------
int kmap_embed_launch_fast_rows(kmap_embed *e, float *G, hipStream_t st);                 // embed_fast.hip
int kmap_embed_launch_sym(kmap_embed *e, float *G, bool reduce_into_G, hipStream_t st);   // embed_fast.hip
int kmap_embed_launch_seq(kmap_embed *e, float *G, hipStream_t st);                       // embed_seq.hip
void kmap_embed_seq_split(kmap_embed *e);                                                 // embed_seq.hip: rows -> pair / quad / wide form
int kmap_embed_seq_blocks(const kmap_embed *e);                                           // blocks (= loss partials) of the SEQ launch
int kmap_embed_seq_blocks_max(const kmap_embed *e);                                       // ... of whichever form a later set_prob may select

inline int kmap_embed_force_blocks(const kmap_embed *e) {
    if (e->sym) return (int)(e->n_lblocks * (((e->symJ + SY_WAVES - 1) / SY_WAVES) * SY_WAVES));
    if (e->mode == KMAP_EMBED_SEQ) return kmap_embed_seq_blocks(e);
    return (int)((e->nrows + F_RPW * F_WAVES - 1) / (F_RPW * F_WAVES));
}

// ---- peer-direct exchange (peer_exchange.hip owns the object; embed.hip's kmap_embed_step_peer drives it) ------------------------
constexpr int KMAP_PEER_MAX = 16;
struct kmap_peer {
    int world = 0, rank = 0;
    int64_t msg_floats = 0;                    // floats of a message (2 N + MSG_EXTRA)
    int64_t slot_floats = 0;                   // ... rounded up to a multiple of 4: slot stride (16-byte copies)
    size_t area_bytes = 0;
    void *area = nullptr;                      // this rank's receive area: slots float[2][world][slot_floats] | flags u64[2][world]
    void *peer_area[KMAP_PEER_MAX] = {};       // every rank's area as mapped here (own entry = area)
    bool opened[KMAP_PEER_MAX] = {};
    float *msg_local = nullptr;                // the message forces_msg writes (entries of other ranks' rows stay zero)
    unsigned long long *done = nullptr;        // push kernel's finished-blocks counter | sticky time-out flag
    uint64_t iter = 0;                         // iterations issued so far
    unsigned long long timeout_ticks = 1000000000ull;   // bound of the apply kernel's wait in 100 MHz wall-clock ticks (10 s)
};
struct PeerTab {                               // kernel argument
    float *slots[KMAP_PEER_MAX];               // base of rank q's slots
    unsigned long long *flags[KMAP_PEER_MAX];  // base of rank q's flags
};
// area layout: slots float[2][world][slot_floats] | flags u64[2][world] | hello u64[world]
inline size_t kmap_peer_slots_bytes(const kmap_peer *p) { return (size_t)2 * p->world * p->slot_floats * 4; }
inline size_t kmap_peer_hello_offset(const kmap_peer *p) { return kmap_peer_slots_bytes(p) + (size_t)2 * p->world * 8; }
