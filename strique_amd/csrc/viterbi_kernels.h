// Internal: profile-HMM Viterbi kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace strq {

// Device image of a baked model (strique_amd/hmm.py: bake), laid out for one wave64.
// Every lane owns up to `epl` emitting and `spl` silent states (slot-major: entry slot*64 + lane).
// With layout hints (strq_model_create) a state sits in the lane of its profile position, so that
// the j-th in-edge of all lanes of a slot reads consecutive LDS cells (no bank conflicts);
// otherwise states are dealt to slots by descending in-degree.
// Silent states are laid out in chains: when the highest-numbered silent predecessor of a silent
// state is free, the pair becomes chain neighbours and that edge leaves the edge list (chain_src /
// chain_logp).  A chain zig-zags through the silent slots: position p sits in lane p / spl, slot p % spl.
// In-edge j of the state owned by (slot, lane) is entry (base[slot] + j) * 64 + lane of
// edge_src / edge_logp; padding edges point at the extra cell v[n_states] == -inf.
// Register-resident layout of a profile chain (viterbi_g2_kernel): the model's states as ONE chain of positions g, two
// positions per lane (g = 2 * lane + parity), every position holding at most a match-type state M_g, an insert-type state
// I_g (both emitting) and a delete-type silent state D_g.  All in-edges of the regular part connect a position with itself
// or its predecessor, so a time step needs the lane's own previous values and those of lane - 1 -- no gathers:
//   M_g <- M_{g-2}, I_{g-1}, M_{g-1}, I_g, M_g, [B0], D_{g-1}             (B0 only at even g)
//   I_g <- I_g, M_g, D_g
//   D_g <- I_{g-1}, M_{g-1}, [B1] (this time step's values; B1 only at even g), then its chain predecessor D_{g-1}
// in this order -- which must be the ascending order of the source states, the order the oracle breaks ties in; B0 / B1 are
// two designated emitting states (STRique: the dummy states that close the repeat loop) whose values are broadcast.
// Round 4: an insert-type state has three columns, not seven.  What bake() splices out of STRique's graph -- the silent
// hubs repeat.e1 (in front of dummy1) and repeat.s1 (in front of the first insert of the repeat unit), scripts/STRique.py:
// 339-344 -- comes back in the IMAGE as a virtual delete-type state at a free position (the repeat unit has no delete
// states): dummy1 <- V <- {last insert, last match of the unit}, and repeat0i <- {itself, repeat0m, V' <- {dummy1, last
// prefix delete}}.  A virtual state forwards with log-probability 0.0: x + lp + 0.0 == x + lp bit for bit, the payload of a
// silent state is its predecessor's, and the relayed sources are the LAST of their target's in-edges in evaluation order
// and adjacent, so that every tie is broken as in the baked model (g2_layout checks all of it and refuses otherwise).
// Rows of `lp` (64 doubles each, -inf where a lane has no such edge): see G2_ROW_* below.
struct VitG2 {
    const double* lp;            // G2_ROWS x 64
    const double* em;            // [slot][a | b | c][lane]: emission parameters of the emitting slots (Me, Mo, Ie, Io), as in VitModel
    const int32_t* kind;         // [slot][lane]: 0 none, 1 Normal, 2 Uniform
    const int32_t* own;          // [6][lane]: state of Me, Mo, Ie, Io, De, Do; -1 none, -2 a virtual relay state
    const int32_t* inc;          // [4][lane]: count_inc of the emitting states
    const int32_t* tag;          // [4][lane]: state_tag == 1
    int32_t bc_slot[2], bc_lane[2];      // broadcast sources B0 (slot 0 / 1: feeds match-type states) and B1 (slot 2 / 3: feeds delete-type states); lane -1: none
    int32_t start_slot, start_lane, end_slot, end_lane;      // silent slots 0 (even g) / 1 (odd g)
    const uint64_t* mark_add;    // [4][lane] or null: what an emission of (slot, lane) adds to the 64-bit payload of a mark decode (see G2_MARK_*)
    uint64_t hub_mask;           // lanes whose even delete slot is a virtual relay that lets its target win a tie when the relay's own winner was one of its gather columns
};
// Payload of a mark decode on this layout (want_bp 2: the modification pass needs the stretch of the window decoded into the
// repeat section, scripts/STRique.py:608): three counters in one 64-bit integer, every emission adds a per-(slot, lane) constant
// with one 64-bit addition -- [0, 21) emissions from tagged states, [21, 43) visits of counted states, [43, 64) emissions from
// untagged states BEHIND the tagged stretch of the chain.  The model is one-way (prefix -> repeat section -> suffix), so with
// T observations the first tagged emission is observation T - behind - tagged and the first one after the section T - behind.
enum { G2_MARK_COUNT_SHIFT = 21, G2_MARK_BEHIND_SHIFT = 43 };
enum { G2_ROW_ME = 0, G2_ROW_MO = 7, G2_ROW_IE = 13, G2_ROW_IO = 16, G2_ROW_DE = 19, G2_ROW_DO = 22, G2_ROW_CHAIN = 24, G2_ROWS = 26 };

struct VitModel {
    int32_t n_states, n_emit, n_silent, start, end;
    int32_t epl, spl;                 // slots per lane (emitting / silent)
    int32_t e_deg[8], e_base[8];      // padded in-degree and first edge row of every emitting slot
    int32_t s_deg[8], s_base[8];
    int32_t n_edge_rows;
    int32_t single_stage;             // 1: no silent state has a silent predecessor outside its chain
    int32_t n_cells, start_cell, end_cell;   // LDS cells: emitting slot s lane l -> s*64+l, silent -> (epl+s)*64+l, last = -inf
    const int32_t* edge_src;          // n_edge_rows * 64: LDS cell of the source state
    const int32_t* cell_state;        // n_cells: state held by a cell, -1 if none
    const double* edge_logp;          // n_edge_rows * 64
    const int32_t* own_e;             // epl * 64: state owned by (slot, lane) or -1
    const int32_t* own_s;             // spl * 64
    const int32_t* chain_src;         // spl * 64: the chain predecessor of the cell -- the state in (slot - 1, lane), for slot 0 in (spl - 1, lane - 1) -- or -1
    const double* chain_logp;         // spl * 64: log-probability of that chain edge
    const int32_t* emis_kind;         // epl * 64, by owner slot  (0 = padding)
    const double* emis_a;             // mu | lo
    const double* emis_b;             // 1/(2 sigma^2) | hi
    const double* emis_c;             // -log(sigma sqrt(2pi)) | -log(hi - lo)
    const int32_t* count_inc;         // n_states + 1, by state
    const int32_t* state_tag;         // n_states + 1, by state
    double uni_lo_max, uni_hi_min;    // tightest bounds of the uniform emissions: observations inside them need no range test
    int32_t rec_state;                // the hub state (tag 2) with an edge into `end` (e0 of the modification model), or -1
    int32_t silent_counted;           // 1: some silent state has a non-zero count_inc (STRique counts emitting states only: dummy1 / dummy2)
    int32_t e_flat[8];                // emitting slot without a Normal emission (uniform inserts, padding): its emission is a constant per lane
    // Models no lane layout covers (more than 512 emitting / 256 silent states, more than 8 in-edges): the baked arrays as
    // they are, for viterbi_csr_kernel -- one workgroup per window, a cell per state, silent states level by level.
    int32_t csr, n_levels;            // csr = 1: the fields above that describe a lane layout are unused (epl = spl = 0, cell = state)
    const int32_t* csr_in_ptr;        // n_states + 1
    const int32_t* csr_in_src;        // in-edges by ascending source
    const double* csr_in_logp;
    const int32_t* csr_kind;          // n_emit: 1 Normal, 2 Uniform
    const double* csr_a;
    const double* csr_b;
    const double* csr_c;
    const int32_t* csr_level_ptr;     // n_levels + 1: silent states by the length of their longest silent predecessor chain
    const int32_t* csr_level_state;   // n_silent
    const VitG2* g2;                  // register-resident profile layout of the same model, or null (strq_model_set_positions)
    int32_t g2_odd, g2_mark;          // that image has its broadcast sources at odd positions; it can carry the repeat-section marks (mark_add)
};
#define VIT_SHAPE_CSR 8              // launch_viterbi shape id of those models
#define VIT_SHAPE_G2 9               // ... of models with a VitG2 image, for count / mark launches (want_bp 0 or 2); both parities (g2_odd) share the launch
#define VIT_CSR_MAX_STATES 4096      // two buffers of 16-byte cells in 160 KB of LDS

enum { VIT_SRC_F64 = 0, VIT_SRC_F64_AFFINE = 1, VIT_SRC_I16_AFFINE = 2 };

struct VitTask {
    const VitModel* model;   // device image of the HMM this window is decoded with
    const void* sig;         // first observation
    int64_t T;
    int32_t src_kind, pad_;
    double c1, h1, h2, c2, lo, hi;    // x = clip((s - c1) / h1 * h2 + c2, lo, hi)  (STRique.py:159-160,178-179)
    uint16_t* bp;            // (T + 1) x (n_states) predecessor states, nullable (count-only mode); hub mode: (T + 1) 8-byte hub records
};

struct VitResult {
    double logp;
    int64_t counted;
    int32_t status;          // 0 ok, 1 no path
    int32_t pad_;
    uint32_t dbg[4];         // reserved (zero)
};

// One launch decodes windows of several models as long as they fit the same kernel shape
// (`shape_of`); `max_cells` = largest n_cells among them (checked against the shape's LDS buffers).
#define VIT_SHAPE_SS 16                            // flag in the shape id: single-stage model
int vit_shape_of(const VitModel& model_host);      // -1 if no compiled shape fits
int vit_shape_for(const VitModel& model_host, int want_bp);      // the same, or VIT_SHAPE_G2 when the model has a register-resident image and the mode allows it
int vit_shape_silent_slots(int shape);             // silent slots per lane of that kernel shape
int launch_viterbi(hipStream_t stream, int shape, int max_cells, const VitTask* tasks, VitResult* results,
                   int n_tasks, int* queue, int n_cu, int want_bp, const int* order = nullptr, int waves_hint = 0);
// waves_hint 4: the launch shares the GPU with other kernels (register-resident shape: four waves per workgroup instead of eight)
// want_bp: 0 = count only, 1 = back-pointers, 2 = repeat-section marks (flanked model), 3 = hub records (modification model)
int launch_vit_sort(hipStream_t stream, const VitTask* tasks, int n, int* order);   // order by descending T (n <= 8192)
int launch_vit_traceback(hipStream_t stream, const VitTask* tasks, const VitResult* results,
                         int32_t* const* paths, int n_tasks);

}  // namespace strq
