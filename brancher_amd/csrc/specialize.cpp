// specialize.cpp — program specialisation: a model program (include/bsvi.h) becomes straight-line HIP for gfx950.
//
// The interpreter of elbo_kernel.hip pays ~170-200 instructions per node visit for fetch, decode and the LDS round
// trips of its memory-to-memory operands; the arithmetic of a Normal node is ~15.  A model is compiled ONCE per
// (joint, posterior, estimator) and then evaluated thousands of times (brancher/inference.py:95-108), so the library
// generates the per-sample body as HIP source — every slot a register, every record loop unrolled, every weight and
// flag a literal — wraps it in the hand-written frame of spec_prelude.h / spec_main.h, and compiles it with hiprtc at
// the first launch.  What the reference re-derives per call in Python (variables.py:486-570,718-749: graph walk,
// name mapping, broadcasting) was already resolved by the lowering; this removes the last interpretive layer.
//
// Semantics are the interpreter's, instruction for instruction (elbo_kernel.hip exec_forward / exec_backward /
// naff_sink and the two sweeps of elbo_block): the parity tests run every fixture through both engines.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <link.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "bsvi.h"
#include "bsvi_internal.h"
#include "spec_args.h"
#include "jit_headers.inc"      // kJitHeaderNames / kJitHeaderTexts: the device headers, embedded by the Makefile

namespace bsvi_spec {

using bsvi::SpecArgs;

// ---------------------------------------------------------------------------------------------------------------
//  the generator
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct Insn { uint32_t w0, dst, a, b, c, s, imm0, imm1; };

constexpr size_t kMaxVisits = 12000;     // unrolled instruction visits (forward + reverse) a body may have
constexpr uint32_t kKeepEpsRows = 64;    // up to this many noise rows stay in registers for the reverse sweep
constexpr uint32_t kRescheduleAboveSlots = 40; // programs with more per-sample slots defer their sinks (register pressure)
constexpr uint32_t kFenceAboveCode = 100;     // programs longer than this get scheduling fences ...
constexpr const char* kJitOptionMark = "// bsvi-jit-option: ";     // a per-program hiprtc option carried in the generated source
constexpr uint32_t kFenceEvery = 4;           // ... every this many records
constexpr uint32_t kBasicRegallocAboveCode = 700;   // programs longer than this are compiled with LLVM's basic register allocator
constexpr uint32_t kAccumulateEntries = 1u << 30;   // up to this many gradient-carrying uniform entries accumulate in registers

std::string fmt(const char* f, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

std::string flit(uint32_t bits) {       // a float literal that round-trips
    float x;
    memcpy(&x, &bits, 4);
    if (!isfinite(x)) return fmt("__uint_as_float(0x%08xu)", bits);
    if (x == (float)(long long)x && fabsf(x) < 1e6f) return fmt("%.1ff", (double)x);
    return fmt("%.9gf", (double)x);
}

class Emitter {
public:
    Emitter(const bsvi_program_desc& d, bool diag) : d_(d), diag_(diag) {
        keep_eps_ = d.n_noise <= kKeepEpsRows;
        // long programs: a gradient contribution leaves through its own position of the transpose tile instead of
        // waiting in an accumulator register for the entry's last contribution
        direct_du_ = d.n_uniform_grad > kAccumulateEntries;
        // (short programs have the registers to spare and run a little faster with every sink in the forward sweep,
        //  where the scheduler finds more independent work: 6.75 vs 7.7 kcycles per body at BASELINE config 1)
        reschedule_ = d.n_slots > kRescheduleAboveSlots;
        for (uint32_t pc = 0; pc < d.n_code; ++pc) {
            const Insn I = insn(pc);
            const uint32_t op = I.w0 & 0xFFu;
            if ((op == BSVI_OP_NAFF || op == BSVI_OP_NODE) && (((I.w0 >> 8) & 0xFFu) & BSVI_F_WF) && d.estimator != BSVI_EST_BLACKBOX) reschedule_ = false;
        }
        // A score term (BlackBox) weights the reverse steps with the COMPLETE f, so the sinks cannot simply move into the
        // reverse sweep.  Long programs run them TWICE instead: value only in the forward sweep (f complete at the turn),
        // adjoints only — same schedule as above — in the reverse sweep.  With every sink finished in the forward sweep the
        // adjoint of every latent is live across the turn: T = 200 is 2 x 201 long-lived registers of 512, and the register
        // allocator's eviction search alone took 23 s of a 29 s compile (-ftime-report).
        two_pass_ = reschedule_ && d.estimator == BSVI_EST_BLACKBOX && !(getenv("BSVI_SPEC_TWO_PASS") && getenv("BSVI_SPEC_TWO_PASS")[0] == '0');
        if (d.estimator == BSVI_EST_BLACKBOX && !two_pass_) reschedule_ = false;
        du_total_.assign(d.n_uniform_grad, 0);
        du_seen_.assign(d.n_uniform_grad, 0);
    }

    bool run(std::string& why) {
        counting_ = true;
        walk();
        if (visits_ > kMaxVisits) { why = fmt("unrolled stream has %zu instruction visits (limit %zu)", visits_, kMaxVisits); return false; }
        counting_ = false;
        body_.clear();
        fwd_groups_.clear(); rev_groups_.clear(); node_rows_.clear();
        walk();
        // entries nothing contributed to still own a position (their gradient is 0)
        for (uint32_t k = 0; k < d_.n_uniform_grad; ++k)
            if (!du_total_[k]) complete(k, "0.0f");
        return true;
    }

    const std::string& body() const { return body_; }
    bool keeps_noise() const { return keep_eps_; }
    // body of spec_draw(): the standard normals of the sample's Normal nodes, one Philox call per group of 4 rows
    std::string draw() const {
        std::string s;
        if (!keep_eps_) return s;
        std::set<uint32_t> groups;
        for (uint32_t r : normal_rows_) groups.insert(r >> 2);
        std::string philox;
        for (uint32_t g : groups) {
            philox += fmt("        { float z0, z1, z2, z3; spec_normals4(A, T, %uu, z0, z1, z2, z3);", g);
            for (uint32_t j = 0; j < 4; ++j)
                if (normal_rows_.count(4 * g + j)) philox += fmt(" Z.z[%u] = z%u;", 4 * g + j, j);
            philox += " }\n";
        }
        if (!diag_) return "    {\n" + philox + "    }\n";
        s += "    if (A.noise) {\n";
        for (uint32_t r : normal_rows_) s += fmt("        Z.z[%u] = A.noise[(size_t)%uu * A.n_local + T.nc];\n", r, r);
        s += "    } else {\n" + philox + "    }\n";
        return s;
    }
    const std::vector<uint32_t>& order() const { return order_; }     // position -> uniform entry
    std::string declarations() const {
        std::string s;
        for (uint32_t i = 0; i < d_.n_slots; ++i) s += fmt("    float v_%u = 0.0f, a_%u = 0.0f;\n", i, i);
        if (!direct_du_) for (uint32_t k = 0; k < d_.n_uniform_grad; ++k) if (du_total_[k]) s += fmt("    float du_%u = 0.0f;\n", k);
        for (uint32_t g : all_groups_) s += fmt("    float pn_%u_0 = 0.0f, pn_%u_1 = 0.0f, pn_%u_2 = 0.0f, pn_%u_3 = 0.0f;\n", g, g, g, g);
        for (uint32_t r : eps_rows_) s += fmt("    float ez_%u = 0.0f;\n", r);
        for (uint32_t r : all_node_rows_) s += fmt("    float ns_%u = 0.0f;\n", r);
        return s;
    }

private:
    const bsvi_program_desc& d_;
    bool diag_;
    bool keep_eps_ = true, counting_ = true, direct_du_ = false, reschedule_ = true, two_pass_ = false;
    int sink_mode_ = 0;         // two_pass_: 1 = a sink's value only (forward sweep), 2 = its adjoints only (reverse sweep); 0 = both
    size_t visits_ = 0;
    std::string body_;
    std::vector<uint32_t> du_total_, du_seen_, order_;
    std::set<uint32_t> fwd_groups_, rev_groups_, all_groups_, eps_rows_, node_rows_, all_node_rows_, normal_rows_;

    Insn insn(uint32_t pc) const {
        Insn I;
        memcpy(&I, d_.code + 8 * (size_t)pc, 32);
        return I;
    }
    void line(const std::string& s) { if (!counting_) { body_ += "    "; body_ += s; body_ += "\n"; } }

    // ---- operands (include/bsvi.h: byte_offset | walks<<30 | per_lane<<31)
    struct Opnd { bool lane; uint32_t idx; };
    static Opnd resolve(uint32_t o, uint32_t e) {
        const bool lane = o >> 31;
        const uint32_t step = ((o >> 30) & 1u) * (lane ? 8u : 4u);
        const uint32_t off = (o & 0x3FFFFFFFu) + e * step;
        return Opnd{lane, lane ? off / 8u : off / 4u};
    }
    std::string val(uint32_t o, uint32_t e) const {
        const Opnd p = resolve(o, e);
        return p.lane ? fmt("v_%u", p.idx) : fmt("SPEC_U(%u)", p.idx);
    }
    void complete(uint32_t k, const std::string& expr) {
        if (counting_) return;
        const uint32_t pos = (uint32_t)order_.size();
        order_.push_back(k);
        line(fmt("SPEC_DU(%uu, %s);", pos, expr.c_str()));
    }
    // scatter an adjoint: slots accumulate in their register; parameter-sourced uniform entries in theirs, leaving
    // through the transpose tile at their last contribution; constants and observed data take none
    void add_adj(uint32_t o, uint32_t e, const std::string& expr) {
        const Opnd p = resolve(o, e);
        if (p.lane) { line(fmt("a_%u += %s;", p.idx, expr.c_str())); return; }
        if (p.idx >= d_.n_uniform_grad) return;
        if (counting_) { ++du_total_[p.idx]; return; }
        if (direct_du_) { complete(p.idx, expr); return; }
        line(fmt("du_%u += %s;", p.idx, expr.c_str()));
        if (++du_seen_[p.idx] == du_total_[p.idx]) complete(p.idx, fmt("du_%u", p.idx));
    }

    // 1/S and log S of a Normal node's scale: companions of the uniform table when S is lane-uniform
    std::string rcp_of(uint32_t o, uint32_t e) const {
        const Opnd p = resolve(o, e);
        return p.lane ? fmt("spec_rcp(v_%u)", p.idx) : fmt("SPEC_UR(%u)", p.idx);
    }
    std::string log_of(uint32_t o, uint32_t e) const {
        const Opnd p = resolve(o, e);
        return p.lane ? fmt("spec_log(v_%u)", p.idx) : fmt("SPEC_UL(%u)", p.idx);
    }

    // ---- noise of a Normal draw.  Programs that keep their noise (keep_eps_) take it from spec_draw's SpecNoise,
    //      drawn ahead of the barrier; longer ones draw group by group inside the sweeps.
    std::string normal_var(uint32_t row, bool reverse) {
        const uint32_t g = row >> 2, j = row & 3u;
        std::set<uint32_t>& have = (reverse && !keep_eps_) ? rev_groups_ : fwd_groups_;
        if (!have.count(g)) {
            have.insert(g);
            all_groups_.insert(g);
            // (the reverse sweep's second draw of a group goes through an opaque copy of the group number: the same call
            //  on the same literal is a common subexpression of the forward sweep's, and the optimiser would keep that
            //  one's four normals in registers across the whole body instead of drawing again)
            const std::string gs = (reverse && !keep_eps_) ? std::string("spec_opaque(") + fmt("%uu", g) + ")" : fmt("%uu", g);
            if (diag_) line(fmt("if (!noise) spec_normals4(A, T, %s, pn_%u_0, pn_%u_1, pn_%u_2, pn_%u_3);", gs.c_str(), g, g, g, g));
            else line(fmt("spec_normals4(A, T, %s, pn_%u_0, pn_%u_1, pn_%u_2, pn_%u_3);", gs.c_str(), g, g, g, g));
        }
        return fmt("pn_%u_%u", g, j);
    }
    std::string eps_forward(uint32_t row) {
        if (keep_eps_) { normal_rows_.insert(row); return fmt("Z.z[%u]", row); }
        const std::string pn = normal_var(row, false);
        if (!diag_) return pn;
        eps_rows_.insert(row);
        line(fmt("ez_%u = noise ? noise[(size_t)%uu * A.n_local + T.nc] : %s;", row, row, pn.c_str()));
        return fmt("ez_%u", row);
    }
    std::string eps_reverse(uint32_t row) {
        if (keep_eps_) return fmt("Z.z[%u]", row);
        const std::string pn = normal_var(row, true);
        if (!diag_) return pn;
        line(fmt("ez_%u = noise ? noise[(size_t)%uu * A.n_local + T.nc] : %s;", row, row, pn.c_str()));
        return fmt("ez_%u", row);
    }
    void diag_outputs(uint32_t row, const std::string& v, const std::string& noise) {
        if (!diag_) return;
        line(fmt("if (T.active) { if (A.samples_out) A.samples_out[(size_t)%uu * A.n_local + T.n] = %s; "
                 "if (A.noise_out) A.noise_out[(size_t)%uu * A.n_local + T.n] = %s; }", row, v.c_str(), row, noise.c_str()));
    }

    // ---- forward of one instruction at element e (elbo_kernel.hip exec_forward)
    void forward(const Insn& I, uint32_t e, bool nodes) {
        const uint32_t op = I.w0 & 0xFFu, flags = (I.w0 >> 8) & 0xFFu, dist = (I.w0 >> 16) & 0xFFu;
        if (op == BSVI_OP_NAFF) {
            if (!nodes) return;
            ++visits_;
            if (counting_) return;
            line("{");
            line(fmt("  const float loc = %s * %s + %s;", val(I.a, e).c_str(), val(I.b, e).c_str(), val(I.c, e).c_str()));
            std::string v = val(I.dst, e);
            if (flags & BSVI_F_SAMPLE) {
                const uint32_t row = resolve(I.dst, e).idx;
                const std::string eps = eps_forward(row);
                if (flags & BSVI_F_GIVEN) line(fmt("  %s = %s;", v.c_str(), eps.c_str()));
                else line(fmt("  %s = loc + %s * %s;", v.c_str(), eps.c_str(), val(I.s, e).c_str()));
                diag_outputs(row, v, eps);
            }
            if (flags & (BSVI_F_ENT | BSVI_F_LOGP | BSVI_F_WF)) line(fmt("  const float lS = %s;", log_of(I.s, e).c_str()));
            if (flags & BSVI_F_ENT) line(fmt("  %s += %s * (kHalfLog2PiE + lS);", ftarget(), flit(I.imm1).c_str()));
            if (flags & (BSVI_F_LOGP | BSVI_F_WF)) {
                line(fmt("  const float lp = spec_naff_lp(%s, loc, %s, lS);", v.c_str(), rcp_of(I.s, e).c_str()));
                line(fmt("  %s += %s * lp;", ftarget(), flit(I.imm0).c_str()));
                if ((flags & BSVI_F_WF) && sink_mode_ != 2) line("  T.lq += lp;");
            }
            line("}");
        } else if (op == BSVI_OP_BIN) {
            ++visits_;
            if (counting_) return;
            const std::string a = val(I.a, e), b = val(I.b, e), y = val(I.dst, e);
            switch (flags) {
            case BSVI_B_ADD: line(fmt("%s = %s + %s;", y.c_str(), a.c_str(), b.c_str())); break;
            case BSVI_B_SUB: line(fmt("%s = %s - %s;", y.c_str(), a.c_str(), b.c_str())); break;
            case BSVI_B_MUL: line(fmt("%s = %s * %s;", y.c_str(), a.c_str(), b.c_str())); break;
            case BSVI_B_DIV: line(fmt("%s = %s / %s;", y.c_str(), a.c_str(), b.c_str())); break;
            case BSVI_B_POW: line(fmt("%s = pow_ff(%s, %s);", y.c_str(), a.c_str(), b.c_str())); break;
            default: line(fmt("%s = (%s == %s) ? 1.0f : 0.0f;", y.c_str(), a.c_str(), b.c_str())); break;
            }
        } else if (op == BSVI_OP_UN) {
            ++visits_;
            if (counting_) return;
            line(fmt("%s = unop<true>(%uu, %s, %s);", val(I.dst, e).c_str(), flags, val(I.a, e).c_str(), flit(I.imm0).c_str()));
        } else if (op == BSVI_OP_NODE) {
            if (!nodes) return;
            ++visits_;
            if (counting_) return;
            line("{");
            line(fmt("  const float p0 = %s, p1 = %s;", val(I.a, e).c_str(), val(I.b, e).c_str()));
            const std::string v = val(I.dst, e);
            if (flags & BSVI_F_SAMPLE) {
                const uint32_t row = resolve(I.dst, e).idx;
                node_rows_.insert(row);
                all_node_rows_.insert(row);
                const std::string draw = fmt("{ const float2 dr = philox_draw(spec_key(A, T), %u, p0, p1, %uu); %s = dr.x; ns_%u = dr.y; }",
                                             dist, row, v.c_str(), row);
                if (diag_) {
                    const std::string given = (flags & BSVI_F_GIVEN) ? fmt("ns_%u", row) : fmt("sample_from_noise_generic(%u, p0, p1, ns_%u)", dist, row);
                    line(fmt("  if (noise) { ns_%u = noise[(size_t)%uu * A.n_local + T.nc]; %s = %s; } else %s", row, row, v.c_str(), given.c_str(), draw.c_str()));
                } else {
                    line("  " + draw);
                }
                diag_outputs(row, v, fmt("ns_%u", row));
            }
            if (flags & BSVI_F_ENT) line(fmt("  %s += %s * entropy_generic(%u, p0, p1);", ftarget(), flit(I.imm1).c_str(), dist));
            if (flags & (BSVI_F_LOGP | BSVI_F_WF)) {
                line(fmt("  const float lp = logp_generic(%u, %s, p0, p1);", dist, v.c_str()));
                line(fmt("  %s += %s * lp;", ftarget(), flit(I.imm0).c_str()));
                if ((flags & BSVI_F_WF) && sink_mode_ != 2) line("  T.lq += lp;");
            }
            line("}");
        }
    }

    // where a term's VALUE goes: the sample's f — or nowhere, when the term runs a second time for its adjoints (two_pass_)
    const char* ftarget() const { return sink_mode_ == 2 ? "f_dead" : "T.f"; }

    // the weight of a term's GRADIENT: the record's constant — times the caller's per-sample weight in the diagnostic variant
    // (bsvi_elbo_args::f_weight_dev: the second pass of a user-defined gradient estimator; 1 without one)
    std::string wg(uint32_t imm) const { return diag_ ? "(" + flit(imm) + " * T.gw)" : flit(imm); }

    // ---- a model log-prob term N(value | A*B + C, S) finished in the forward sweep (elbo_kernel.hip naff_sink)
    void naff_sink(const Insn& I, uint32_t e) {
        ++visits_;
        const std::string A = val(I.a, e), B = val(I.b, e);
        if (sink_mode_ == 1) {      // the value alone
            line(fmt("T.f += %s * spec_naff_lp(%s, %s * %s + %s, %s, %s);", flit(I.imm0).c_str(), val(I.dst, e).c_str(), A.c_str(), B.c_str(),
                     val(I.c, e).c_str(), rcp_of(I.s, e).c_str(), log_of(I.s, e).c_str()));
            return;
        }
        if (!counting_) {
            line("{");
            line(fmt("  float gl, gs; spec_naff_sink(%s, %s, %s * %s + %s, %s, %s, %s, gl, gs);", flit(I.imm0).c_str(), val(I.dst, e).c_str(),
                     A.c_str(), B.c_str(), val(I.c, e).c_str(), rcp_of(I.s, e).c_str(), log_of(I.s, e).c_str(), ftarget()));
            if (diag_) line("  gl *= T.gw; gs *= T.gw;");
        }
        add_adj(I.dst, e, "-gl");
        add_adj(I.a, e, "gl * " + B);
        add_adj(I.b, e, "gl * " + A);
        add_adj(I.c, e, "gl");
        add_adj(I.s, e, "gs");
        line("}");
    }

    // ---- reverse of one instruction at element e (elbo_kernel.hip exec_backward)
    void backward(const Insn& I, uint32_t e) {
        const uint32_t op = I.w0 & 0xFFu, flags = (I.w0 >> 8) & 0xFFu, dist = (I.w0 >> 16) & 0xFFu;
        ++visits_;
        if (op == BSVI_OP_NAFF) {
            const std::string A = val(I.a, e), B = val(I.b, e), v = val(I.dst, e);
            if (!counting_) {
                line("{");
                line(fmt("  const float rS = %s, loc = %s * %s + %s;", rcp_of(I.s, e).c_str(), A.c_str(), B.c_str(), val(I.c, e).c_str()));
                line("  float gv = 0.0f, gl = 0.0f, gs = 0.0f;");
                if (flags & (BSVI_F_LOGP | BSVI_F_WF)) {
                    const std::string gw = (flags & BSVI_F_WF) ? fmt("(%s + fweight)", wg(I.imm0).c_str()) : wg(I.imm0);
                    line(fmt("  spec_naff_lp_bwd(%s, %s, loc, rS, gv, gl, gs);", gw.c_str(), v.c_str()));
                }
                if (flags & BSVI_F_ENT) line(fmt("  gs += %s * rS;", wg(I.imm1).c_str()));
            }
            if (flags & BSVI_F_SAMPLE) {
                // a sampled latent's own adjoint is its incoming gradient: folded into loc / scale, left unchanged
                if (!counting_) {
                    const Opnd dst = resolve(I.dst, e);
                    const std::string eps = eps_reverse(dst.idx);
                    line(fmt("  const float zb = a_%u + gv;", dst.idx));
                    line("  gl += zb;");
                    line(fmt("  gs += zb * %s;", eps.c_str()));
                }
            } else {
                add_adj(I.dst, e, "gv");
            }
            add_adj(I.a, e, "gl * " + B);
            add_adj(I.b, e, "gl * " + A);
            add_adj(I.c, e, "gl");
            add_adj(I.s, e, "gs");
            line("}");
        } else if (op == BSVI_OP_BIN) {
            const std::string a = val(I.a, e), b = val(I.b, e);
            const Opnd dst = resolve(I.dst, e);
            const std::string g = fmt("a_%u", dst.idx), y = fmt("v_%u", dst.idx);
            switch (flags) {
            case BSVI_B_ADD: add_adj(I.a, e, g); add_adj(I.b, e, g); break;
            case BSVI_B_SUB: add_adj(I.a, e, g); add_adj(I.b, e, "-" + g); break;
            case BSVI_B_MUL: add_adj(I.a, e, g + " * " + b); add_adj(I.b, e, g + " * " + a); break;
            case BSVI_B_DIV:
                line(fmt("{ const float ga = %s / %s;", g.c_str(), b.c_str()));
                add_adj(I.a, e, "ga");
                add_adj(I.b, e, "-ga * " + y);
                line("}");
                break;
            case BSVI_B_POW:
                line(fmt("{ const float g = %s, pa = %s, pb = %s;", g.c_str(), a.c_str(), b.c_str()));
                add_adj(I.a, e, "g * pb * pow_ff(pa, pb - 1.0f)");
                add_adj(I.b, e, "((g == 0.0f) ? 0.0f : g * " + y + " * logf(pa))");
                line("}");
                break;
            default: break;
            }
        } else if (op == BSVI_OP_UN) {
            const Opnd dst = resolve(I.dst, e);
            const std::string a = val(I.a, e);
            add_adj(I.a, e, fmt("a_%u * unop_grad<true>(%uu, %s, v_%u, %s)", dst.idx, flags, a.c_str(), dst.idx, flit(I.imm0).c_str()));
        } else if (op == BSVI_OP_NODE) {
            const std::string v = val(I.dst, e);
            if (!counting_) {
                line("{");
                line(fmt("  const float p0 = %s, p1 = %s;", val(I.a, e).c_str(), val(I.b, e).c_str()));
                line("  float gv = 0.0f, g0 = 0.0f, g1 = 0.0f;");
                if (flags & (BSVI_F_LOGP | BSVI_F_WF)) {
                    const std::string gw = (flags & BSVI_F_WF) ? fmt("(%s + fweight)", wg(I.imm0).c_str()) : wg(I.imm0);
                    line(fmt("  { const float4 r = logp_bwd_generic(%u, %s, p0, p1, %s); gv += r.x; g0 += r.y; g1 += r.z; }", dist, v.c_str(), gw.c_str()));
                }
                if (flags & BSVI_F_ENT)
                    line(fmt("  { const float2 r = entropy_bwd_generic(%u, p0, p1, %s); g0 += r.x; g1 += r.y; }", dist, wg(I.imm1).c_str()));
            }
            if (flags & BSVI_F_SAMPLE) {
                if (!counting_) {
                    const Opnd dst = resolve(I.dst, e);
                    const bool base = dist == BSVI_DIST_LOGNORMAL || dist == BSVI_DIST_CAUCHY || dist == BSVI_DIST_LAPLACE;
                    const std::string noise = base ? fmt("ns_%u", dst.idx) : v;
                    line(fmt("  { const float2 r = sample_bwd_generic(%u, %s, p0, p1, %s, a_%u + gv); g0 += r.x; g1 += r.y; }",
                             dist, v.c_str(), noise.c_str(), dst.idx));
                }
            } else {
                add_adj(I.dst, e, "gv");
            }
            add_adj(I.a, e, "g0");
            add_adj(I.b, e, "g1");
            line("}");
        }
    }

    void zero_temps(uint32_t base, uint32_t n) {
        for (uint32_t t = 0; t < n; ++t) line(fmt("a_%u = 0.0f;", base + t));
    }

    // ---- the records of the stream
    struct Item { uint32_t first, n, n_elems, temp_base, n_temps; bool bracket, sink; };
    std::vector<Item> items() const {
        std::vector<Item> v;
        for (uint32_t pc = 0; pc < d_.n_code;) {
            const Insn I = insn(pc);
            const bool sink = (I.w0 >> 24) & BSVI_R_SINK;
            if ((I.w0 & 0xFFu) == BSVI_OP_REC_BEGIN) {
                v.push_back(Item{pc + 1, I.dst, I.a, I.b, I.c, true, sink});
                pc += I.dst + 2;
            } else {
                v.push_back(Item{pc, 1, 1, 0, 0, false, sink});
                pc += 1;
            }
        }
        return v;
    }
    void emit_forward(const Item& R) {          // forward sweep of a record; a sink also runs its reverse step here
        if (!R.bracket) {
            const Insn I = insn(R.first);
            if (R.sink && (I.w0 & 0xFFu) == BSVI_OP_NAFF) { naff_sink(I, 0); return; }
            forward(I, 0, true);
            if (R.sink && sink_mode_ != 1) backward(I, 0);
            return;
        }
        for (uint32_t e = 0; e < R.n_elems && visits_ <= kMaxVisits; ++e) {
            for (uint32_t j = 0; j < R.n; ++j) forward(insn(R.first + j), e, true);
            if (R.sink && sink_mode_ != 1) {
                zero_temps(R.temp_base, R.n_temps);
                for (uint32_t j = R.n; j-- > 0;) backward(insn(R.first + j), e);
            }
        }
    }
    void emit_reverse(const Item& R) {          // reverse sweep of a (non-sink) record
        if (!R.bracket) { backward(insn(R.first), 0); return; }
        for (uint32_t e = R.n_elems; e-- > 0 && visits_ <= kMaxVisits;) {
            for (uint32_t j = 0; j < R.n; ++j) forward(insn(R.first + j), e, false);      // re-materialise the temps
            zero_temps(R.temp_base, R.n_temps);
            for (uint32_t j = R.n; j-- > 0;) backward(insn(R.first + j), e);
        }
    }
    // slots an instruction reads or scatters adjoints to / the slot it defines, over the record's elements
    void slots_of(const Item& R, std::set<uint32_t>& reads, std::set<uint32_t>& writes) const {
        for (uint32_t j = 0; j < R.n; ++j) {
            const Insn I = insn(R.first + j);
            const uint32_t op = I.w0 & 0xFFu, flags = (I.w0 >> 8) & 0xFFu;
            const uint32_t ops[5] = {I.dst, I.a, I.b, I.c, I.s};
            const int n_opnd = (op == BSVI_OP_NAFF) ? 5 : ((op == BSVI_OP_UN) ? 2 : 3);
            const bool defines = op == BSVI_OP_BIN || op == BSVI_OP_UN || (flags & BSVI_F_SAMPLE);
            for (uint32_t e = 0; e < R.n_elems; ++e)
                for (int k = 0; k < n_opnd; ++k) {
                    const Opnd p = resolve(ops[k], e);
                    if (!p.lane) continue;
                    if (k == 0 && defines) writes.insert(p.idx); else reads.insert(p.idx);
                }
        }
    }

    // ---- the two sweeps of elbo_block.  The interpreter runs every sink record (a model log-prob term: value AND
    //      adjoints in one visit) in the forward sweep, so the adjoint of every latent stays live from there to the
    //      latent's turn in the reverse sweep — 2 registers per latent across the whole body.  Here a sink is deferred to
    //      just before the reverse step of the LAST-defined slot it touches (the first of them the reverse sweep
    //      consumes): adjoints live for a few instructions.  Not with a score term in the program (BlackBox: the reverse
    //      steps weight log q with the COMPLETE f).
    void walk() {
        visits_ = 0;
        const std::vector<Item> recs = items();
        // Reverse order of the non-sink records.  Reverse stream order is one valid order; any order works in which a
        // record's reverse step comes after the reverse steps of the records that read what it defines.  A DERIVED value
        // (an arithmetic record, e.g. the sigmoid(b) that every transition prior of an AR chain reads) is defined late
        // in the stream, so in reverse stream order it would come first and pull every sink that reads it to the front.
        // It is anchored instead to the latest-defined latent among its inputs: its reverse step runs just before that
        // latent's.  key = (anchor, index), visited in descending order.
        std::vector<std::pair<int, int>> key(recs.size(), std::make_pair(-1, -1));
        std::vector<int> def_of(d_.n_slots, -1);
        std::vector<uint32_t> rorder;                     // non-sink records in reverse-sweep order
        for (uint32_t r = 0; r < recs.size(); ++r) {
            if (recs[r].sink) continue;
            std::set<uint32_t> rd, wr;
            slots_of(recs[r], rd, wr);
            bool samples = false;
            for (uint32_t j = 0; j < recs[r].n; ++j) {
                const Insn I = insn(recs[r].first + j);
                const uint32_t op = I.w0 & 0xFFu;
                if ((op == BSVI_OP_NAFF || op == BSVI_OP_NODE) && (((I.w0 >> 8) & 0xFFu) & BSVI_F_SAMPLE)) samples = true;
            }
            int anchor = (int)r;
            if (!samples && reschedule_) {
                int a = -1;
                for (uint32_t sl : rd) {
                    if (sl >= recs[r].temp_base && sl < recs[r].temp_base + recs[r].n_temps) continue;
                    if (def_of[sl] >= 0 && key[def_of[sl]].first > a) a = key[def_of[sl]].first;
                }
                if (a >= 0) anchor = a;
            }
            key[r] = std::make_pair(anchor, (int)r);
            for (uint32_t sl : wr) def_of[sl] = (int)r;
            rorder.push_back(r);
        }
        std::sort(rorder.begin(), rorder.end(), [&](uint32_t x, uint32_t y) { return key[x] > key[y]; });
        std::vector<std::vector<uint32_t>> deferred(recs.size());
        std::vector<int> home(recs.size(), -1);
        if (reschedule_) {
            for (uint32_t r = 0; r < recs.size(); ++r) {
                if (!recs[r].sink) continue;
                std::set<uint32_t> rd, wr;
                slots_of(recs[r], rd, wr);
                int first = -1;                           // the definer of its slots that the reverse sweep visits first
                for (uint32_t sl : rd) {
                    if (sl >= recs[r].temp_base && sl < recs[r].temp_base + recs[r].n_temps) continue;     // its own temps
                    const int dr = def_of[sl];         // (latents and derived values are defined once)
                    if (dr >= 0 && (uint32_t)dr < r && (first < 0 || key[dr] > key[first])) first = dr;
                }
                if (first >= 0) { home[r] = first; deferred[first].push_back(r); }
            }
        }
        // long programs: a compiler fence every few records.  A node's uniform entries (scale, 1/scale, log scale) are read
        // in the forward sweep and again in the reverse sweep; with no store in between that provably aliases them the
        // optimiser keeps the first read's value instead of reading LDS again — three registers per node held across
        // the whole body (T = 200: 600 registers, spilled to scratch).  The fence makes it read again.
        const bool fence = d_.n_code > kFenceAboveCode;
        uint32_t since = 0;
        auto maybe_fence = [&]() { if (fence && ++since >= kFenceEvery) { since = 0; line("asm volatile(\"\" ::: \"memory\"); __builtin_amdgcn_sched_barrier(0);"); } };
        if (two_pass_) line("float f_dead = 0.0f; (void)f_dead;");
        for (uint32_t r = 0; r < recs.size(); ++r) {
            if (home[r] >= 0 && !two_pass_) continue;
            sink_mode_ = home[r] >= 0 ? 1 : 0;
            emit_forward(recs[r]);
            sink_mode_ = 0;
            maybe_fence();
            if (visits_ > kMaxVisits) return;
        }
        // the weight of grad log q_n (BlackBox: the complete f_n; a caller's q_weight in the diagnostic variant)
        if (diag_) line("const float fweight = A.q_weight ? A.q_weight[T.nc] : T.f * T.gw; (void)fweight;");
        else line("const float fweight = T.f; (void)fweight;");
        if (fence) {
            // long programs: the reverse sweep recomputes a node's location from its parents' values; as a common
            // subexpression of the forward sweep's the optimiser would instead keep all of them (one register per node
            // across the body).  Passing the slot values through an empty asm makes them new values to it.
            line("asm volatile(\"\" ::: \"memory\");");
            for (uint32_t s0 = 0; s0 < d_.n_slots; s0 += 16) {
                std::string ops;
                for (uint32_t sl = s0; sl < d_.n_slots && sl < s0 + 16; ++sl) ops += fmt("%s\"+v\"(v_%u)", sl == s0 ? "" : ", ", sl);
                line("asm volatile(\"\" : " + ops + ");");
            }
        }
        for (uint32_t r : rorder) {
            sink_mode_ = two_pass_ ? 2 : 0;
            for (uint32_t sidx : deferred[r]) emit_forward(recs[sidx]);
            sink_mode_ = 0;
            emit_reverse(recs[r]);
            maybe_fence();
            if (visits_ > kMaxVisits) return;
        }
        if (diag_) line("if (A.fvalue_out && T.active) { A.fvalue_out[T.n] = T.f; A.fvalue_out[(size_t)A.n_local + T.n] = T.lq; }");
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
//  a program's specialisation: sources, compiled modules, device tables
// ---------------------------------------------------------------------------------------------------------------
struct Variant {
    std::string src;
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    bool failed = false;
    int per_cu = 0;          // workgroups of this code object a CU holds (occupancy query, once)
};

// two launch geometries, each with its own compiled kernels (the transpose-tile size is part of the generated body):
//   ONE  a shard of up to max_threads / 64 waves runs as ONE workgroup — no grid reduction, the training loop can stay
//        in the kernel;
//   MANY larger shards: 256-thread workgroups, sized so that two fit a CU's LDS, each walking several sample chunks.
enum { GEOM_ONE = 0, GEOM_MANY = 1 };
struct Geom { uint32_t max_threads = 0, lds_bytes = 0; };

struct Spec {
    uint32_t n_params = 0, n_uniform = 0, n_ugrad = 0, n_obs = 0, n_noise = 0;
    uint32_t n_pos = 0;                                   // positions of the transpose tile (>= n_ugrad)
    Geom geom[2];
    std::vector<uint32_t> pu_ptr_host, pu_pos_host, pu_idx_host;   // CSR theta -> (position, uniform entry)
    Variant variant[6];                                   // [geometry][0 lean, 1 diagnostic]; 4: lean one-workgroup kernel with the draw wave;
                                                          // 5: the same with the cross-rank exchange inside the training loop (spec_main.h)
    bool exchange_ok = false;                             // every parameter has its owner in one wave: the in-loop exchange serves
    void* dev = nullptr;                                  // [tickets: 256 B][pu_ptr][pu_pos][pu_idx]
    unsigned int* tickets = nullptr;
    const uint32_t* pu_ptr = nullptr;
    const uint32_t* pu_pos = nullptr;
    const uint32_t* pu_idx = nullptr;
    uint32_t n_cus = 256;
    bool draw_wave_ok = false;                            // the in-kernel loop may be given a wave that draws for the owners' wave
    uint32_t launch_seq = 0;
    std::mutex mu;
};

// throughput regime (many workgroups): workgroups of 256 threads per CU the kernel is compiled for — 2: the whole register
// file for two waves per SIMD; 3 / 4: the launch bound is raised to 768 / 1024 threads so that a lane gets 168 / 128
// registers (spilling the rest) and contributions leave through DPP row sums, not the 17 KB-per-wave transpose tile
static uint32_t many_waves() {
    static const uint32_t w = [] { const char* e = getenv("BSVI_SPEC_MANY_WAVES"); const int v = e ? atoi(e) : 2; return (uint32_t)(v < 2 ? 2 : v > 4 ? 4 : v); }();
    return w;
}
static bool tiled(uint32_t n_pos, int geom) {     // SPEC_TILE (spec_prelude.h, SPEC_DU)
    static const bool many_tile = [] { const char* e = getenv("BSVI_SPEC_MANY_TILE"); return !(e && e[0] == '0'); }();
    if (geom == 1 /* GEOM_MANY */ && (!many_tile || many_waves() > 2)) return false;
    return n_pos <= 64;
}
static uint32_t lds_floats(uint32_t n_params, uint32_t n_uniform, uint32_t n_obs, uint32_t n_pos, uint32_t max_threads, int geom) {
    // mirrors the SPEC_OFF_* layout of spec_prelude.h
    const uint32_t W = max_threads / 64;
    const uint32_t u_pad = (n_uniform + n_obs + 3) / 4 * 4, nu_pad = (n_uniform + 3) / 4 * 4;
    const uint32_t ws_pad = tiled(n_pos, geom) ? (n_pos + 3) / 4 * 4 + 4 : 4 * n_pos + 4;
    const uint32_t np_pad = (n_params + 3) / 4 * 4 + 4, tab = (4 * n_uniform + (2 * n_params + 1) + 2 * n_pos + 3) / 4 * 4 + 4;
    const uint32_t own = 20 * (n_params < max_threads ? n_params : max_threads), scr = 4 * (n_pos + 2) + 4;     // SPEC_OWN_WORDS
    return u_pad + 2 * nu_pad + W * ws_pad + (2 * W + 8) + 5 * np_pad + tab + own + scr + (tiled(n_pos, geom) ? W * 64 * 68 : 0);
}

Spec* create(const bsvi_program_desc& d, std::string& why) {
    if (!d.n_code) { why = "empty program"; return nullptr; }
    Spec* s = new Spec();
    s->n_params = d.n_params; s->n_uniform = d.n_uniform; s->n_ugrad = d.n_uniform_grad; s->n_obs = d.n_obs; s->n_noise = d.n_noise;
    {
        // the positions (and with them the CSR map theta -> positions) do not depend on the geometry
        Emitter E(d, false);
        if (!E.run(why)) { delete s; return nullptr; }
        const std::vector<uint32_t>& order = E.order();
        s->n_pos = (uint32_t)order.size();
        std::vector<std::vector<uint32_t>> pos_of(d.n_uniform_grad);
        for (uint32_t p = 0; p < order.size(); ++p) pos_of[order[p]].push_back(p);
        s->pu_ptr_host.assign(1, 0u);
        for (uint32_t i = 0; i < d.n_params; ++i) {
            for (uint32_t j = d.param_uniform_ptr[i]; j < d.param_uniform_ptr[i + 1]; ++j)
                for (uint32_t p : pos_of[d.param_uniform_idx[j]]) { s->pu_pos_host.push_back(p); s->pu_idx_host.push_back(d.param_uniform_idx[j]); }
            s->pu_ptr_host.push_back((uint32_t)s->pu_pos_host.size());
        }
    }
    // launch bounds: a sample keeps its slot values and its noise in registers; 512 threads leave 256 registers per
    // lane, 256 threads the whole 512-entry file (MI355X_MICROARCH.md, register files)
    const uint32_t live = 2 * d.n_slots + (d.n_noise <= kKeepEpsRows ? d.n_noise : 0) + (d.n_uniform_grad <= kAccumulateEntries ? d.n_uniform_grad : 0) + 40;
    auto fit = [&](uint32_t threads, Geom& g) {
        g.max_threads = threads;
        g.lds_bytes = lds_floats(d.n_params, d.n_uniform, d.n_obs, s->n_pos, threads, (int)(&g - s->geom)) * 4u;
        return g.lds_bytes <= 160u * 1024u;
    };
    bool ok = (live <= 232 && fit(512, s->geom[GEOM_ONE])) || fit(256, s->geom[GEOM_ONE]);
    ok = ok && fit(256, s->geom[GEOM_MANY]);
    if (!ok) { why = "the program's tables do not fit LDS"; delete s; return nullptr; }
    bool all_fast = d.n_params <= 64;      // every parameter owned by a thread of the smallest workgroup, <= 2 positions
    for (uint32_t i = 0; i < d.n_params && all_fast; ++i) all_fast = s->pu_ptr_host[i + 1] - s->pu_ptr_host[i] <= 2;
    // (spec_main.h, SPEC_DRAW_WAVE: the extra wave hands the normals over through its own, unused, transpose tile)
    s->draw_wave_ok = all_fast && tiled(s->n_pos, GEOM_ONE) && d.n_noise > 0 && d.n_noise <= kKeepEpsRows;
    uint32_t ut_mask = 0;                  // the transforms the program's uniform table uses
    for (uint32_t k = 0; k < d.n_uniform; ++k) ut_mask |= 1u << (d.uniform[k].transform & 31u);
    const bool rare_transforms = (ut_mask >> (BSVI_UT_SIGMOID + 1)) != 0;
    for (int gi = 0; gi < 2; ++gi) {
        const Geom& G = s->geom[gi];
        for (int v = 0; v < 2; ++v) {
            Emitter E(d, v == 1);
            if (!E.run(why)) { delete s; return nullptr; }
            std::string src;
            src += "// generated by libbsvi (specialize.cpp) from a model program: do not edit\n";
            src += "#define BSVI_SPECIALIZED 1\n";
            // long BlackBox programs: no SLP vectorizer.  gfx950 has packed f32 arithmetic, so the vectorizer pairs the
            // isomorphic terms of DISTANT records of the unrolled stream into <2 x float> operations placed at the later one
            // — every such pair holds the earlier record's operands across the records in between (T = 200: 2 158 spilled
            // registers and 52 s of compile time with it, 231 and 29 s without; the rest went with the two-pass sinks: 12
            // and 9 s; 199 -> 90 us per iteration at 1 024 samples).  Long Pathwise programs keep it: their deferred sinks
            // leave it nothing distant to pair (7 spills with, 0 without, and 69.7 against 70.9 us).
            // (BSVI_JIT_SLP=1 / 0 forces it on / off for every program: measurements)
            const char* const slp = getenv("BSVI_JIT_SLP");
            if (slp ? slp[0] == '0' : (d.n_code > kFenceAboveCode && d.estimator == BSVI_EST_BLACKBOX))
                src += std::string(kJitOptionMark) + "-fno-slp-vectorize\n";
            // very long programs: the greedy allocator's time grows much faster than the unrolled stream once the register file is
            // full (a Gaussian process over 100 inputs, 414 instructions: 15 s, of which 9 s are "Greedy Register Allocator"; over 200
            // inputs, 814 instructions: 150 s and 690 spilled registers).  The basic allocator takes 16 s there (1 875 spills —
            // beside the term's batched factorisation the program is a small part of such an iteration); same arithmetic.
            // (BSVI_JIT_REGALLOC=greedy / basic forces one for every program)
            const char* const ra = getenv("BSVI_JIT_REGALLOC");
            if (ra ? ra[0] == 'b' : d.n_code > kBasicRegallocAboveCode)
                src += std::string(kJitOptionMark) + "-mllvm\n" + kJitOptionMark + "-vgpr-regalloc=basic\n";
            src += fmt("#define SPEC_N_PARAMS %u\n#define SPEC_N_UNIFORM %u\n#define SPEC_N_UGRAD %u\n#define SPEC_N_OBS %u\n#define SPEC_N_NOISE %u\n",
                       d.n_params, d.n_uniform, d.n_uniform_grad, d.n_obs, d.n_noise);
            src += fmt("#define SPEC_N_POS %u\n", s->n_pos);
            src += fmt("#define SPEC_ESTIMATOR %u\n#define SPEC_MAX_THREADS %u\n#define SPEC_DIAG %d\n", d.estimator, G.max_threads, v);
            src += fmt("#define SPEC_ACCUMULATE_CHUNKS %d\n#define SPEC_TILE %d\n", gi == GEOM_MANY ? 1 : 0, tiled(s->n_pos, gi) ? 1 : 0);
            if (gi == GEOM_MANY && many_waves() > 2) src += fmt("#define SPEC_BOUND_THREADS %u\n", 256u * many_waves());
            src += fmt("#define SPEC_KEEP_NOISE %u\n", E.keeps_noise() ? d.n_noise : 0u);
            // (all parameters "fast": the epilogue's generic loop over the LDS working copy is compiled out)
            src += fmt("#define SPEC_GENERIC_OWNERS %d\n", all_fast ? 0 : 1);
            // (the out-of-line transforms — exp, log, tanh, sqrt, square — are calls: programs without them compile none)
            src += fmt("#define SPEC_RARE_TRANSFORMS %d\n#define SPEC_UT_MASK 0x%xu\n", rare_transforms ? 1 : 0, ut_mask);
            src += "#include \"spec_prelude.h\"\n";
            src += "namespace bsvi {\n";
            src += "__device__ __forceinline__ void spec_draw(const SpecBody& A, const SpecLane& T, SpecNoise& Z) {\n";
            src += "    (void)A; (void)T; (void)Z;\n";
            src += E.draw();
            src += "}\n";
            src += "__device__ __forceinline__ void spec_body(const SpecBody& A, SpecLane& T, const SpecNoise& Z, float* WSw) {\n";
            src += "    (void)Z;\n";
            if (v == 1) src += "    const float* const noise = A.noise;\n";
            src += E.declarations();
            src += E.body();
            src += "}\n}  // namespace bsvi\n";
            src += "#include \"spec_main.h\"\n";
            s->variant[2 * gi + v].src = std::move(src);
        }
    }
    // (a kernel of its own: the extra roles cost the plain loop 3 % when they are merely compiled in)
    s->variant[4].src = "#define SPEC_WITH_DRAW_WAVE 1\n" + s->variant[0].src;
    s->exchange_ok = true;       // (owners in one wave: the wave exchanges; otherwise every thread exchanges its own parameters' entries)
    s->variant[5].src = std::string("#define SPEC_WITH_EXCHANGE 1\n") + (s->draw_wave_ok ? "#define SPEC_WITH_DRAW_WAVE 1\n" : "") + s->variant[0].src;
    return s;
}

void destroy(Spec* s) {
    if (!s) return;
    for (Variant& v : s->variant) if (v.module) (void)hipModuleUnload(v.module);
    if (s->dev) (void)hipFree(s->dev);
    delete s;
}

int upload(Spec* s) {
    const size_t n_ptr = s->pu_ptr_host.size(), n_pos = s->pu_pos_host.size();
    const size_t bytes = 512 + (n_ptr + 2 * n_pos) * 4 + 256;      // [arrival tickets: 64 words][generation numbers: 64 words][CSR tables]
    hipError_t e = hipMalloc(&s->dev, bytes);
    if (e != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, std::string("hipMalloc (specialiser tables): ") + hipGetErrorString(e));
    std::vector<uint32_t> host(128 + n_ptr + 2 * n_pos, 0u);
    std::copy(s->pu_ptr_host.begin(), s->pu_ptr_host.end(), host.begin() + 128);
    std::copy(s->pu_pos_host.begin(), s->pu_pos_host.end(), host.begin() + 128 + n_ptr);
    std::copy(s->pu_idx_host.begin(), s->pu_idx_host.end(), host.begin() + 128 + n_ptr + n_pos);
    e = hipMemset(s->dev, 0, bytes);
    if (e == hipSuccess) e = hipMemcpy(s->dev, host.data(), host.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, std::string("specialiser tables: ") + hipGetErrorString(e));
    s->tickets = (unsigned int*)s->dev;
    s->pu_ptr = (const uint32_t*)s->dev + 128;
    s->pu_pos = s->pu_ptr + n_ptr;
    s->pu_idx = s->pu_pos + n_pos;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        s->n_cus = (uint32_t)cus;
    return BSVI_OK;
}

// variant: 0 training kernel, 1 diagnostic kernel of the one-workgroup geometry; 2, 3 the same of the many-workgroup one
const std::string& source(const Spec* s, int variant) { return s->variant[(variant == 4 || variant == 5) ? variant : (variant & 3)].src; }

// ---------------------------------------------------------------------------------------------------------------
//  hiprtc
// ---------------------------------------------------------------------------------------------------------------
// (no packed-f32 VALU instructions: see the Makefile)
static const char* kJitOptions[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"};
static const int kJitOptionCount = (int)(sizeof(kJitOptions) / sizeof(kJitOptions[0]));

int compile(const std::string& src, std::vector<char>& code, std::string& log) {
    hiprtcProgram prog = nullptr;
    hiprtcResult r = hiprtcCreateProgram(&prog, src.c_str(), "bsvi_spec.hip", kJitHeaderCount, kJitHeaderTexts, kJitHeaderNames);
    if (r != HIPRTC_SUCCESS) { log = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r); return BSVI_ERR_HIP; }
    // per-program options ride in the source as "// bsvi-jit-option: <opt>" lines (so the disk cache's key, which hashes
    // the source and the common array, covers them)
    std::vector<std::string> extra;
    for (size_t at = src.find(kJitOptionMark); at != std::string::npos; at = src.find(kJitOptionMark, at + 1)) {
        const size_t b = at + strlen(kJitOptionMark), e = src.find('\n', b);
        extra.push_back(src.substr(b, e == std::string::npos ? std::string::npos : e - b));
    }
    std::vector<const char*> opts(kJitOptions, kJitOptions + kJitOptionCount);
    for (const std::string& o : extra) opts.push_back(o.c_str());
    r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    size_t n = 0;
    if (hiprtcGetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) {
        log.resize(n);
        (void)hiprtcGetProgramLog(prog, &log[0]);
    }
    if (r != HIPRTC_SUCCESS) {
        log = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(r) + "\n" + log;
        (void)hiprtcDestroyProgram(&prog);
        return BSVI_ERR_HIP;
    }
    size_t cs = 0;
    r = hiprtcGetCodeSize(prog, &cs);
    if (r == HIPRTC_SUCCESS) { code.resize(cs); r = hiprtcGetCode(prog, code.data()); }
    (void)hiprtcDestroyProgram(&prog);
    if (r != HIPRTC_SUCCESS) { log = std::string("hiprtcGetCode: ") + hiprtcGetErrorString(r); return BSVI_ERR_HIP; }
    if (const char* dump = getenv("BSVI_JIT_DUMP")) {       // tools: keep the code object for llvm-objdump / llvm-readelf
        if (FILE* f = fopen(dump, "wb")) { fwrite(code.data(), 1, code.size(), f); fclose(f); }
    }
    return BSVI_OK;
}

// code objects of this process, by source text: programs lowered twice (tests, estimator siblings) compile once
static std::mutex g_cache_mu;
static std::map<std::string, std::vector<char>> g_code_cache;

// ---------------------------------------------------------------------------------------------------------------
//  code objects on disk: hiprtc costs ~1 s per program (4 s at T = 200) in EVERY process that trains a model; the
//  generated source is a pure function of the model program, so a second process loads the code object instead.
//  Key = hash of (generated source, embedded device headers, compile options, target, hiprtc + HIP runtime version);
//  directory = $BSVI_CACHE_DIR, else $XDG_CACHE_HOME/brancher_amd/jit, else $HOME/.cache/brancher_amd/jit, else
//  /tmp/brancher_amd-<uid>/jit.  BSVI_JIT_CACHE=0 switches it off.  Files are written to a temporary name and renamed
//  (atomic on one file system: concurrent ranks compiling the same program race harmlessly); a file carries its own
//  length and content hash, a truncated or foreign one is ignored and overwritten.
// ---------------------------------------------------------------------------------------------------------------
namespace disk_cache {

struct Hash128 { uint64_t a = 0xcbf29ce484222325ull, b = 0x9ae16a3b2f90404full; };
static void feed(Hash128& h, const void* data, size_t n) {
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; ++i) {
        h.a = (h.a ^ p[i]) * 0x100000001b3ull;                                   // FNV-1a
        h.b = (h.b + p[i] + 0x9e3779b97f4a7c15ull) * 0xff51afd7ed558ccdull;      // an independent multiply-xorshift chain
        h.b ^= h.b >> 29;
    }
}
static void feed(Hash128& h, const std::string& s) { const uint64_t n = s.size(); feed(h, &n, 8); feed(h, s.data(), s.size()); }

static bool enabled() {
    const char* e = getenv("BSVI_JIT_CACHE");
    return !(e && e[0] == '0');
}

static std::string directory() {
    if (const char* d = getenv("BSVI_CACHE_DIR")) if (d[0]) return d;
    if (const char* x = getenv("XDG_CACHE_HOME")) if (x[0]) return std::string(x) + "/brancher_amd/jit";
    if (const char* h = getenv("HOME")) if (h[0]) return std::string(h) + "/.cache/brancher_amd/jit";
    return fmt("/tmp/brancher_amd-%u/jit", (unsigned)getuid());
}

static bool make_dirs(const std::string& path) {
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') {
            const std::string part = path.substr(0, i);
            if (mkdir(part.c_str(), 0700) != 0 && errno != EEXIST) return false;
        }
    return true;
}

// The cache holds GPU code that this process will run: it is only trusted when the directory is a real directory (not a
// link) that belongs to this user and that nobody else can write.  The fallback path under /tmp is predictable, the key is
// a non-cryptographic hash of known text and the embedded content hash authenticates the file, not its origin — another
// local user who got there first could otherwise plant a code object.  Anything else: no disk cache (hiprtc compiles).
static bool trusted(const std::string& dir) {
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0) return false;
    return S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}

// WHICH compiler will run.  hiprtc hands the translation unit to libamd_comgr (clang + lld inside), and which libamd_comgr a
// process has is not a property of the hiprtc it calls: a Python host that imported torch first runs the ROCm libraries bundled
// with the torch wheel (roc-7.0.2 here: clang 20), the same host under rocprofv3 — whose tool library brings the system's
// libamd_comgr.so.3 in first — runs /opt/rocm's (roc-7.2.0: clang 22), and the two produce different code for the same source
// and options (BASELINE config 1's loop kernel: 4 938 against 5 283 instructions, 40 against 62 spilled scalar registers;
// profiles/r5/jit_cache_root_cause.txt).  Round 4's key hashed hiprtcVersion / hipRuntimeGetVersion only, so code objects stored
// by profiled runs were served to unprofiled processes, which then compared them bit for bit with variants they compiled
// themselves: the one-ulp failures of the suite after tools/r4/profile_r4.sh.  The key now carries the identity of every loaded
// module that can take part in a compilation — path, size and modification time of each libamd_comgr / libhiprtc /
// libLLVM / libclang in load order (the order decides which definition a symbol binds to).  When no libamd_comgr is loaded yet
// (a host whose hiprtc opens it lazily) the identity is UNRESOLVED: nothing is loaded to find out (round 5 opened it by soname, which
// could itself decide which copy the process ends up with — ADVICE r5); such a process compiles its first kernel with hiprtc, has
// its compiler loaded from then on, and stores and looks up under the resolved identity.  The identity is read again at every key.
static int identity_cb(struct dl_phdr_info* info, size_t, void* data) {
    const char* name = info->dlpi_name;
    if (!name || !*name) return 0;
    const char* base = strrchr(name, '/');
    base = base ? base + 1 : name;
    const char* const parts[] = {"libamd_comgr", "libhiprtc", "libLLVM", "libclang-cpp", "libclang.so"};
    bool is_part = false;
    for (const char* p : parts) is_part = is_part || strncmp(base, p, strlen(p)) == 0;
    if (!is_part) return 0;
    struct stat st;
    std::string& id = *(std::string*)data;
    id += name;
    if (stat(name, &st) == 0) id += fmt(":%lld:%lld", (long long)st.st_size, (long long)st.st_mtime);
    id += ";";
    return 0;
}
std::string compiler_identity() {
    std::string s;
    dl_iterate_phdr(identity_cb, &s);
    return s;
}
static bool identity_resolved(const std::string& id) { return id.find("amd_comgr") != std::string::npos; }

static std::string key_of(const std::string& src) {
    static const Hash128 base = [] {        // everything but the generated source: the same for every program of a process
        Hash128 h;
        for (int i = 0; i < kJitHeaderCount; ++i) { feed(h, std::string(kJitHeaderNames[i])); feed(h, std::string(kJitHeaderTexts[i])); }
        for (int i = 0; i < kJitOptionCount; ++i) feed(h, std::string(kJitOptions[i]));
        int major = 0, minor = 0, runtime = 0;
        (void)hiprtcVersion(&major, &minor);
        (void)hipRuntimeGetVersion(&runtime);
        feed(h, fmt("hiprtc %d.%d runtime %d abi %d", major, minor, runtime, BSVI_ABI_VERSION));
        return h;
    }();
    Hash128 h = base;
    feed(h, compiler_identity());
    feed(h, src);
    return fmt("%016llx%016llx", (unsigned long long)h.a, (unsigned long long)h.b);
}

struct Header { char magic[8]; uint64_t size, ha, hb; };
static const char kMagic[8] = {'B', 'S', 'V', 'I', 'C', 'O', '0', '1'};

static bool load(const std::string& key, std::vector<char>& code) {
    const std::string dir = directory();
    if (!trusted(dir)) return false;
    const std::string path = dir + "/" + key + ".co";
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    FILE* f = fd >= 0 ? fdopen(fd, "rb") : nullptr;
    if (!f) { if (fd >= 0) close(fd); return false; }
    Header H;
    bool ok = fread(&H, sizeof H, 1, f) == 1 && memcmp(H.magic, kMagic, 8) == 0 && H.size > 0 && H.size < (1ull << 30);
    if (ok) {
        code.resize(H.size);
        ok = fread(code.data(), 1, H.size, f) == H.size && fgetc(f) == EOF;
    }
    fclose(f);
    if (ok) {
        Hash128 h;
        feed(h, code.data(), code.size());
        ok = h.a == H.ha && h.b == H.hb;
    }
    if (!ok) code.clear();
    return ok;
}

static void store(const std::string& key, const std::vector<char>& code) {
    const std::string dir = directory();
    if (!make_dirs(dir) || !trusted(dir)) return;
    Header H;
    memcpy(H.magic, kMagic, 8);
    H.size = code.size();
    Hash128 h;
    feed(h, code.data(), code.size());
    H.ha = h.a; H.hb = h.b;
    const std::string tmp = dir + "/" + key + fmt(".tmp.%d", (int)getpid());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
    if (!f) { if (fd >= 0) close(fd); return; }
    const bool ok = fwrite(&H, sizeof H, 1, f) == 1 && fwrite(code.data(), 1, code.size(), f) == code.size();
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), (dir + "/" + key + ".co").c_str()) != 0) (void)remove(tmp.c_str());
}

}  // namespace disk_cache

// what the last ensure_compiled of this thread did (bsvi_jit_cache_stats): tests and tools read it
static thread_local int t_last_source = 0;      // 0 none yet, 1 hiprtc, 2 process cache, 3 disk cache

// the code object of a translation unit: this process's cache, the disk cache, or hiprtc (and then both caches)
int obtain(const std::string& text, std::vector<char>& code, std::string& log, int* origin) {
    {
        std::lock_guard<std::mutex> g(g_cache_mu);
        auto it = g_code_cache.find(text);
        if (it != g_code_cache.end()) code = it->second;
    }
    int from = 2;
    if (code.empty()) {
        std::string src = text;
        const bool tuned = getenv("BSVI_SPEC_DEFINES") != nullptr;
        if (tuned) src = std::string(getenv("BSVI_SPEC_DEFINES")) + "\n" + src;     // tools: timing experiments
        const bool on_disk = disk_cache::enabled() && !getenv("BSVI_JIT_DUMP");
        // (no compiler loaded yet: no lookup — the key would not say whose code it names; the store below is keyed after the compilation)
        const bool lookup = on_disk && disk_cache::identity_resolved(disk_cache::compiler_identity());
        if (lookup && disk_cache::load(disk_cache::key_of(src), code)) {
            from = 3;
        } else {
            const int rc = compile(src, code, log);
            if (rc) return rc;
            from = 1;
            if (on_disk && disk_cache::identity_resolved(disk_cache::compiler_identity())) disk_cache::store(disk_cache::key_of(src), code);
        }
        std::lock_guard<std::mutex> g(g_cache_mu);
        if (!tuned) g_code_cache[text] = code;
    }
    t_last_source = from;
    if (origin) *origin = from;
    return BSVI_OK;
}

int last_origin() { return t_last_source; }
std::string cache_directory() { return disk_cache::enabled() ? disk_cache::directory() : std::string(); }
std::string compiler_identity() { return disk_cache::compiler_identity(); }

static int ensure_compiled(Spec* s, int v) {
    Variant& V = s->variant[v];
    if (V.fn) return BSVI_OK;
    if (V.failed) return BSVI_ERR_UNSUPPORTED;
    std::vector<char> code;
    std::string log;
    if (obtain(V.src, code, log, nullptr)) {
        V.failed = true;
        if (getenv("BSVI_DEBUG")) fprintf(stderr, "bsvi: specialised kernel did not compile:\n%s\n", log.c_str());
        return bsvi_fail(BSVI_ERR_UNSUPPORTED, "specialised kernel did not compile: " + log.substr(0, 400));
    }
    hipError_t e = hipModuleLoadData(&V.module, code.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&V.fn, V.module, "bsvi_spec_kernel");
    if (e != hipSuccess) {
        V.failed = true;
        V.fn = nullptr;
        return bsvi_fail(BSVI_ERR_HIP, std::string("loading the specialised kernel: ") + hipGetErrorString(e));
    }
    return BSVI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
//  launch
// ---------------------------------------------------------------------------------------------------------------
struct Geo { uint32_t blocks, threads; int geom; bool draw_wave = false; uint32_t extra_waves = 0; };
// The in-kernel loop is given one wave more than the samples need: it carries no samples and draws the next iteration's
// normals of the owners' wave, whose chain — draw, body, sums, epilogue — is what an iteration takes (spec_main.h).
static bool draw_wave() {
    const char* e = getenv("BSVI_SPEC_DRAW_WAVE");        // (read per call: the tests switch it within a process)
    return !(e && e[0] == '0');
}
static bool draw_service() {
    const char* e = getenv("BSVI_SPEC_DRAW_SERVICE");     // (read per call: the tests switch it within a process)
    return !(e && e[0] == '0');
}
static Geo geo(const Spec* s, uint32_t n_local, int mode = MODE_SUMS, bool exchange = false) {
    const uint32_t waves = (n_local + 63) / 64;
    if (waves <= s->geom[GEOM_ONE].max_threads / 64) {
        // (up to four sample waves — one per SIMD: beyond that two of them share a SIMD and their draws set the pace, not the owners' chain)
        const bool extra = mode == MODE_LOOP && s->draw_wave_ok && draw_wave() && waves <= 4 && waves + 1 <= s->geom[GEOM_ONE].max_threads / 64
                           && !s->variant[exchange ? 5 : 4].failed;
        // four or five sample waves: four / three draw waves draw for ALL of them (spec_main.h, the draw service; BSVI_SPEC_DRAW_SERVICE=0:
        // the single draw wave up to four sample waves, none at five); the sets are handed over in the draw waves' transpose tiles.
        // (Three sample waves: 3.89 us with the single draw wave, 3.94 with the service; one and two: the single draw wave.)
        const uint32_t n_service = waves == 4 ? 4u : 3u;
        const bool service = mode == MODE_LOOP && s->draw_wave_ok && draw_wave() && draw_service() && waves >= 4 && waves <= 5
                             && waves + n_service <= s->geom[GEOM_ONE].max_threads / 64 && (size_t)s->n_noise * 64u * waves <= (size_t)n_service * 64u * 68u
                             && !s->variant[exchange ? 5 : 4].failed;
        const uint32_t more = service ? n_service : extra ? 1u : 0u;
        return Geo{1, (waves + more) * 64, GEOM_ONE, more != 0u, more};
    }
    // many samples: 256-thread workgroups (one wave per SIMD), two per CU at most; beyond that every workgroup walks
    // several chunks of 256 samples
    const uint32_t threads = s->geom[GEOM_MANY].max_threads;
    uint32_t blocks = (n_local + threads - 1) / threads;
    uint32_t per_cu = many_waves();                                                            // (registers: see many_waves)
    while (per_cu > 1u && s->geom[GEOM_MANY].lds_bytes * per_cu > 160u * 1024u) --per_cu;
    if (blocks > per_cu * s->n_cus) blocks = per_cu * s->n_cus;
    return Geo{blocks, threads, GEOM_MANY};
}

void geometry(const Spec* s, uint32_t n_local, int mode, uint32_t* n_blocks, uint32_t* n_threads, uint32_t* lds_bytes) {
    const Geo g = geo(s, n_local, mode);
    if (n_blocks) *n_blocks = g.blocks;
    if (n_threads) *n_threads = g.threads;
    if (lds_bytes) *lds_bytes = s->geom[g.geom].lds_bytes;
}

size_t workspace_bytes(const Spec* s, uint32_t n_local) {
    const Geo g = geo(s, n_local);
    return ((size_t)g.blocks * (2 + s->n_pos) * 4 + 255) / 256 * 256 + 256;
}

bool applies(const Spec* s, uint32_t n_local, int mode) {
    if (!s || !n_local) return false;
    const char* e = getenv("BSVI_JIT");
    if (e && e[0] == '0') return false;
    const Geo g = geo(s, n_local);
    if (s->variant[2 * g.geom].failed || s->variant[2 * g.geom + 1].failed) return false;
    // several workgroups in loop mode: workgroup 0 owns the iteration, the others wait on its generation number (spec_main.h);
    // every workgroup must be resident — geo() never asks for more than fit the chip (BSVI_SPEC_LOOP_MANY=0: launch per iteration)
    if (mode == MODE_LOOP && g.blocks != 1) {
        const char* m = getenv("BSVI_SPEC_LOOP_MANY");
        return !(m && m[0] == '0');
    }
    return true;
}

int launch(Spec* s, const bsvi_program* p, const Launch& L) {
    const bsvi_elbo_args* a = L.a;
    if (!a->params_dev && s->n_params) return bsvi_fail(BSVI_ERR_INVALID, "params_dev is null");
    if (!a->obs_dev && s->n_obs) return bsvi_fail(BSVI_ERR_INVALID, "obs_dev is null");
    if (!a->out_dev) return bsvi_fail(BSVI_ERR_INVALID, "out_dev is null");
    if (!a->n_samples_local || !a->n_samples_global) return bsvi_fail(BSVI_ERR_INVALID, "zero samples");
    Geo g = geo(s, a->n_samples_local, L.mode, L.xchg != nullptr);
    int v = 2 * g.geom + ((a->noise_dev || a->samples_out_dev || a->noise_out_dev || a->fvalue_out_dev || a->f_weight_dev || a->q_weight_dev) ? 1 : 0);
    if (L.xchg && !(L.mode == MODE_LOOP && g.blocks == 1 && v == 0 && s->exchange_ok))
        return bsvi_fail(BSVI_ERR_UNSUPPORTED, "the in-loop exchange serves the one-workgroup training loop with Philox noise and no per-sample outputs");
    uint32_t seq;
    {
        std::lock_guard<std::mutex> lock(s->mu);
        if (L.xchg) {
            const int rc5 = ensure_compiled(s, 5);
            if (rc5) return rc5;
            v = 5;
        }
        else if (g.draw_wave && v == 0 && ensure_compiled(s, 4) == BSVI_OK) v = 4;
        else if (g.draw_wave) { g.threads -= 64 * g.extra_waves; g.draw_wave = false; }        // (diagnostic kernel, or the variant did not compile)
        const int rc = ensure_compiled(s, v);
        if (rc) return rc;
        seq = s->launch_seq++;
    }
    if (g.blocks > 1 && !L.workspace) return bsvi_fail(BSVI_ERR_INVALID, "workspace_dev is null");
    SpecArgs A;
    memset(&A, 0, sizeof A);
    A.uniform = p->uniform; A.consts = p->consts;
    A.params = L.params ? L.params : const_cast<float*>(a->params_dev);
    A.obs = a->obs_dev; A.noise = a->noise_dev;
    A.samples_out = a->samples_out_dev; A.noise_out = a->noise_out_dev; A.fvalue_out = a->fvalue_out_dev;
    A.out = a->out_dev;
    A.partials = (float*)L.workspace;
    A.ticket = s->tickets + (seq % 64u);     // library-owned, zero between launches
    A.pu_ptr = s->pu_ptr; A.pu_pos = s->pu_pos; A.pu_idx = s->pu_idx;
    A.state = L.state; A.mask = L.mask; A.mask_first = L.mask_first ? L.mask_first : L.mask;
    A.loss_slot = L.loss_slot; A.finite_slot = L.finite_slot;
    A.n_local = a->n_samples_local; A.n_global = a->n_samples_global; A.sample_base = a->sample_base;
    A.mode = (uint32_t)L.mode;
    A.seed_lo = (uint32_t)a->seed; A.seed_hi = (uint32_t)(a->seed >> 32);
    A.offset_lo = (uint32_t)a->offset; A.offset_hi = (uint32_t)(a->offset >> 32);
    A.offset_dev = (const unsigned long long*)a->offset_dev;
    A.f_weight = a->f_weight_dev; A.q_weight = a->q_weight_dev;
    A.xchg = (const bsvi::SpecExchange*)L.xchg;
    A.n_iterations = L.n_iterations; A.pretraining_iterations = L.pretraining_iterations; A.n_params = s->n_params;
    if (L.cfg) A.cfg = *L.cfg;
    if (g.blocks > 1) {
        // every workgroup of the in-kernel loop over several workgroups must be resident (workgroup 0 waits for the others in every
        // iteration): never more than the occupancy of THIS code object allows — fewer workgroups walk more chunks each
        int per_cu = s->variant[v].per_cu;
        if (!per_cu) {
            if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, s->variant[v].fn, (int)g.threads, 0) != hipSuccess || per_cu <= 0) {
                (void)hipGetLastError();
                per_cu = -1;                 // (no answer: geo()'s register / LDS estimate stands)
            }
            s->variant[v].per_cu = per_cu;
        }
        if (per_cu > 0 && g.blocks > (uint32_t)per_cu * s->n_cus) g.blocks = (uint32_t)per_cu * s->n_cus;
        if (L.mode == MODE_LOOP) {
            // the launch's arrival ticket and generation number start from zero (spec_main.h): a launch that gave up on a workgroup
            // leaves the generation at "over", and a straggler may have touched the ticket after workgroup 0 cleared it
            hipError_t me = hipMemsetAsync(A.ticket, 0, 4, (hipStream_t)a->stream);
            if (me == hipSuccess) me = hipMemsetAsync(A.ticket + 64, 0, 4, (hipStream_t)a->stream);
            if (me != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, std::string("hipMemsetAsync (loop ticket): ") + hipGetErrorString(me));
        }
    }
    size_t size = sizeof A;
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    const hipError_t e = hipModuleLaunchKernel(s->variant[v].fn, g.blocks, 1, 1, g.threads, 1, 1, 0, (hipStream_t)a->stream, nullptr, config);
    if (e != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, std::string("hipModuleLaunchKernel (specialised kernel): ") + hipGetErrorString(e));
    return BSVI_OK;
}

}  // namespace bsvi_spec
