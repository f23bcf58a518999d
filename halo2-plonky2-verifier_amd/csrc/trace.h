// trace.h — the op tape an eager context records in trace mode (h2w_ctx_trace_begin): one entry per level-1 / level-2 call of include/h2w.h,
// operands and results named by the CELL OFFSETS their h2w_assigned_t handles carry (every wire of the reference is a cell of Context::advice:
// field/native.rs:28-193 returns AssignedValue handles).  h2w_plan_from_trace (replay.hip) lowers the tape to a device program that
// h2w_fri_witness_batch replays on other proofs of the same shape: the reference's circuit driven ONCE through the operator API, then served at
// GPU speed - no hand restatement of the gadget in between (VERDICT r03 task 4).
// What makes a run replayable: control flow is data-independent, every value that depends on the proof enters as a TAGGED proof word
// (h2w_trace_input) or is computed inside the library (hint-bearing ops are level 2: h2w_gl_div, h2w_gl_ext_inv_witness), and constants are the
// shape's.  The caller's own reads of AssignedValue::value() (the reference's hint sites: base.rs:349,382; extension.rs:327) are what must go
// through the library instead.
#pragma once
#include <string>
#include <vector>
#include <cstdint>
#include "field.h"

namespace h2w {

enum : uint16_t {
    TR_SCOPE_PUSH = 1, TR_SCOPE_POP,
    TR_LOAD_CONSTANT, TR_LOAD_WITNESS, TR_ADD, TR_MUL, TR_MUL_ADD, TR_SELECT, TR_IDX_TO_INDICATOR, TR_SELECT_BY_INDICATOR,
    TR_NUM_TO_BITS, TR_BITS_TO_NUM, TR_DECOMPOSE_LE, TR_LIMBS_TO_NUM, TR_RANGE_CHECK, TR_CLT_SAFE,
    TR_GL_WITNESS, TR_GL_REDUCE, TR_GLOP
};
// where a loaded value comes from: kind 1: proof words [word, word + n); 2: the hint a / b in Goldilocks (base.rs:371-393); 3, 4: component 0, 1 of
// the inverse of the extension element (a, b) (extension.rs:320-340).  a, b: cell offsets.
struct TraceTag { int kind = 0; uint64_t word = 0; uint32_t n = 0; uint64_t a = 0, b = 0; };
struct TraceIn { uint64_t v; uint32_t lit; };      // lit = 0: the cell offset of an operand handle; 1: a literal 64-bit value (a template's fixed operand)
struct TraceOp {
    uint16_t code, sub;              // sub: record template of TR_GLOP
    uint32_t n_in, n_out, first_in, first_out;
    uint64_t imm;                    // constant-pool index / bit count / limb parameters / scope name
    uint64_t cell0; uint32_t ncells; // the cells the call appended: [cell0, cell0 + ncells)
    TraceTag tag;
};
struct Trace {
    std::vector<TraceOp> ops; std::vector<TraceIn> ins; std::vector<uint64_t> outs; std::vector<fr_t> consts; std::vector<std::string> names;
    TraceTag pending; std::string err;
    void fail(const std::string &m) { if (err.empty()) err = m; }
};

}  // namespace h2w
