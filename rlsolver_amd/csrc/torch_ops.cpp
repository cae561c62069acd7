// torch.ops.rlsolver_hip.* -- every DEVICE entry point of include/rlsolver_hip.h as a native PyTorch custom op
// (north_star: "hand-written HIP kernels through PyTorch-ROCm custom ops over a thin C-ABI").
//
// Thin by construction: an op checks device / dtype / contiguity AND every shape against the graph handle and the other
// arguments (a wrongly shaped tensor is a TORCH_CHECK here, never a device out-of-bounds access), switches to the
// tensors' device, takes torch's CURRENT HIP stream there and calls the C-ABI function of the same name in
// librlsolver_hip.so.  This is the one host path of the package: rlsolver_amd/ops*.py, envs/ and methods/ call these ops.  Outputs are caller-allocated and mutated in place, exactly
// like the C ABI (allocation conveniences live in rlsolver_amd/ops.py).  Only the HIP dispatch key gets an
// implementation ("CUDA" is what PyTorch-ROCm calls it): CPU tensors end in the dispatcher's NotImplementedError.
// The shared graph and the spin-system env travel as integer handles = addresses of the host structs rls_graph /
// rls_spin_env (op schemas cannot carry a struct of device pointers).
//
// No kernels here: compiled with the host compiler against libtorch, linked against librlsolver_hip.so.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <optional>

#include "rlsolver_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

inline void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

inline void ok(int rc, const char* fn) {
    TORCH_CHECK(rc == RLS_OK, fn, " failed (", rc, "): ", rls_last_error_string());
}

inline const rls_graph* G(int64_t handle) {
    TORCH_CHECK(handle != 0, "rlsolver_hip: null graph handle");
    return reinterpret_cast<const rls_graph*>(handle);
}

inline void dev(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), name, " must live on a HIP device; rlsolver_hip has no CPU path");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
inline void dev(const Tensor& t, const char* name, at::ScalarType dt) {
    dev(t, name);
    TORCH_CHECK(t.scalar_type() == dt, name, " has dtype ", t.scalar_type(), ", expected ", dt);
}
// shape helpers: every message names the argument
inline void shape2(const Tensor& t, const char* name, int64_t d0, int64_t d1) {
    TORCH_CHECK(t.dim() == 2 && t.size(0) == d0 && t.size(1) == d1, name, " must be [", d0, ", ", d1, "], got ", t.sizes());
}
inline void shape3(const Tensor& t, const char* name, int64_t d0, int64_t d1, int64_t d2) {
    TORCH_CHECK(t.dim() == 3 && t.size(0) == d0 && t.size(1) == d1 && t.size(2) == d2, name, " must be [", d0, ", ", d1, ", ", d2, "], got ",
                t.sizes());
}
inline void count(const Tensor& t, const char* name, int64_t n) {
    TORCH_CHECK(t.numel() == n, name, " must hold ", n, " entries, got ", t.sizes());
}
inline void at_least(const Tensor& t, const char* name, int64_t n) {
    TORCH_CHECK(t.numel() >= n, name, " must hold at least ", n, " entries, got ", t.sizes());
}
inline void same_device(const Tensor& a, const Tensor& b, const char* name) {
    TORCH_CHECK(a.device() == b.device(), name, " is on ", b.device(), ", expected ", a.device());
}
// [B, N] spins against the graph: -> B
inline int64_t env_rows(const Tensor& t, const char* name, const rls_graph* g) {
    TORCH_CHECK(t.dim() == 2 && t.size(1) == g->num_nodes, name, " must be [B, ", g->num_nodes, "], got ", t.sizes());
    return t.size(0);
}
#define RLS_GUARD(t) c10::DeviceGuard rls_device_guard((t).device())
inline void* p(const Tensor& t) { return t.data_ptr(); }
inline void* p(const OptTensor& t) { return t.has_value() ? t->data_ptr() : nullptr; }
inline void optdev(const OptTensor& t, const char* name, at::ScalarType dt) {
    if (t.has_value()) dev(*t, name, dt);
}
// spins: bool / uint8 (1 byte) or float32 (4 bytes); -> spin_bytes
inline int spin_bytes(const Tensor& t, const char* name, bool allow_f32) {
    dev(t, name);
    if (t.scalar_type() == at::kBool || t.scalar_type() == at::kByte) return 1;
    TORCH_CHECK(allow_f32 && t.scalar_type() == at::kFloat, name, " must be bool / uint8", allow_f32 ? " / float32" : "");
    return 4;
}
// chain batches: float32 / uint8 / bool node-major, or int64 = bit-packed tiles (spin_bytes 0)
inline int chain_bytes(const Tensor& t, const char* name) {
    dev(t, name);
    if (t.scalar_type() == at::kLong) return 0;
    if (t.scalar_type() == at::kFloat) return 4;
    TORCH_CHECK(t.scalar_type() == at::kBool || t.scalar_type() == at::kByte, name, " must be float32 / uint8 / bool / int64 (bit-packed)");
    return 1;
}

constexpr auto I64 = at::kLong;
constexpr auto I32 = at::kInt;
constexpr auto F32 = at::kFloat;
constexpr auto F64 = at::kDouble;
constexpr auto U8 = at::kByte;

// ------------------------------------------------------------------------------------------------ MaxCut
void maxcut_obj(int64_t g, const Tensor& xs, Tensor obj) {
    const int sb = spin_bytes(xs, "xs", true);
    const int64_t B = env_rows(xs, "xs", G(g));
    dev(obj, "obj", I64);
    count(obj, "obj", B);
    RLS_GUARD(xs);
    ok(rls_maxcut_obj(G(g), p(xs), sb, B, (int64_t*)p(obj), cur_stream(xs)), "rls_maxcut_obj");
}
void maxcut_edge_cut_mask(int64_t g, const Tensor& xs, Tensor mask) {
    spin_bytes(xs, "xs", false);
    const int64_t B = env_rows(xs, "xs", G(g));
    spin_bytes(mask, "mask", false);
    shape2(mask, "mask", B, G(g)->num_stored_edges);
    RLS_GUARD(xs);
    ok(rls_maxcut_edge_cut_mask(G(g), (const uint8_t*)p(xs), B, (uint8_t*)p(mask), cur_stream(xs)), "rls_maxcut_edge_cut_mask");
}
void maxcut_node_cutdeg(int64_t g, const Tensor& xs, Tensor out) {
    spin_bytes(xs, "xs", false);
    const int64_t B = env_rows(xs, "xs", G(g));
    dev(out, "out", I64);
    shape2(out, "out", B, G(g)->num_nodes);
    RLS_GUARD(xs);
    ok(rls_maxcut_node_cutdeg(G(g), (const uint8_t*)p(xs), B, (int64_t*)p(out), cur_stream(xs)), "rls_maxcut_node_cutdeg");
}
void maxcut_delta_all(int64_t g, const Tensor& xs, Tensor out) {
    spin_bytes(xs, "xs", false);
    const int64_t B = env_rows(xs, "xs", G(g));
    dev(out, "out", I32);
    shape2(out, "out", B, G(g)->num_nodes);
    RLS_GUARD(xs);
    ok(rls_maxcut_delta_all(G(g), (const uint8_t*)p(xs), B, (int32_t*)p(out), cur_stream(xs)), "rls_maxcut_delta_all");
}
void maxcut_step(int64_t g, const Tensor& x_in, Tensor x_out, const Tensor& action, Tensor obj, Tensor reward,
                 const OptTensor& cur, const OptTensor& done, double done_value) {
    const int sb = spin_bytes(x_in, "x_in", true);
    const int64_t B = env_rows(x_in, "x_in", G(g));
    TORCH_CHECK(spin_bytes(x_out, "x_out", true) == sb && x_out.sizes() == x_in.sizes(), "x_in and x_out must have the same shape and dtype");
    dev(action, "action", I64);
    dev(obj, "obj", I32);
    dev(reward, "reward", F32);
    optdev(cur, "cur", F32);
    optdev(done, "done", F32);
    count(action, "action", B);
    count(obj, "obj", B);
    count(reward, "reward", B);
    if (cur.has_value()) count(*cur, "cur", B);
    if (done.has_value()) count(*done, "done", B);
    RLS_GUARD(x_in);
    ok(rls_maxcut_step(G(g), p(x_in), p(x_out), sb, B, (const int64_t*)p(action), (int32_t*)p(obj), (float*)p(reward),
                       (float*)p(cur), (float*)p(done), (float)done_value, cur_stream(x_in)), "rls_maxcut_step");
}
void maxcut_greedy_sweep(int64_t g, Tensor xs, Tensor obj) {
    spin_bytes(xs, "xs", false);
    const int64_t B = env_rows(xs, "xs", G(g));
    dev(obj, "obj", I64);
    count(obj, "obj", B);
    RLS_GUARD(xs);
    ok(rls_maxcut_greedy_sweep(G(g), (uint8_t*)p(xs), B, (int64_t*)p(obj), cur_stream(xs)), "rls_maxcut_greedy_sweep");
}
void maxcut_propose_accept(int64_t g, Tensor xs, const Tensor& mask, Tensor obj) {
    spin_bytes(xs, "xs", false);
    const int64_t B = env_rows(xs, "xs", G(g));
    const bool packed = mask.scalar_type() == I64;      // uint64 bit patterns [ceil(B / 64), N]: the mask as a bit tile
    if (packed) {
        dev(mask, "mask", I64);
        shape2(mask, "mask", (B + 63) / 64, G(g)->num_nodes);
    } else {
        spin_bytes(mask, "mask", false);
        shape2(mask, "mask", B, G(g)->num_nodes);
    }
    dev(obj, "obj", I64);
    count(obj, "obj", B);
    RLS_GUARD(xs);
    ok(rls_maxcut_propose_accept(G(g), (uint8_t*)p(xs), B, p(mask), packed ? 1 : 0, (int64_t*)p(obj), cur_stream(xs)),
       "rls_maxcut_propose_accept");
}
// local-search weights: int8 / int16 / int32 -> ws_bytes
inline int ws_bytes_of(const Tensor& ws, bool allow_i32) {
    dev(ws, "ws");
    if (ws.scalar_type() == at::kChar) return 1;
    if (ws.scalar_type() == at::kShort) return 2;
    TORCH_CHECK(allow_i32 && ws.scalar_type() == I32, "ws must be int8 / int16", allow_i32 ? " / int32" : "", ", got ", ws.scalar_type());
    return 4;
}
void maxcut_ls_weights(int64_t g, const Tensor& xs, int64_t mult, Tensor ws, const OptTensor& ws_minmax) {
    spin_bytes(xs, "xs", false);
    const int64_t B = env_rows(xs, "xs", G(g));
    const int wb = ws_bytes_of(ws, true);
    // ws [B, pitch >= N] contiguous: entries N .. pitch of a row are padding (the round kernels want rows on a 16-byte pitch)
    TORCH_CHECK(ws.dim() == 2 && ws.size(0) == B && ws.size(1) >= G(g)->num_nodes, "ws must be [", B, ", >= ", G(g)->num_nodes, "], got ",
                ws.sizes());
    optdev(ws_minmax, "ws_minmax", I32);
    if (ws_minmax.has_value()) shape2(*ws_minmax, "ws_minmax", 2, G(g)->num_nodes);
    RLS_GUARD(xs);
    ok(rls_maxcut_ls_weights(G(g), (const uint8_t*)p(xs), B, (int32_t)mult, p(ws), wb, ws.size(1), (int32_t*)p(ws_minmax), cur_stream(xs)),
       "rls_maxcut_ls_weights");
}
void maxcut_local_search(int64_t g, Tensor xs, const Tensor& ws, const Tensor& rd_std, const OptTensor& noise, int64_t seed,
                         int64_t env_offset, int64_t num_iters, int64_t num_spin, bool first_draw_proposes, Tensor obj, bool compute_obj) {
    spin_bytes(xs, "xs", false);
    const int64_t B = env_rows(xs, "xs", G(g)), N = G(g)->num_nodes;
    const int wb = ws_bytes_of(ws, false);
    // rows of ws sit ws.size(1) entries apart; the library refuses a pitch that is not a multiple of 16 bytes (it reads 16-byte pieces)
    TORCH_CHECK(ws.dim() == 2 && ws.size(0) == B && ws.size(1) >= N, "ws must be [", B, ", >= ", N, "], got ", ws.sizes());
    dev(rd_std, "rd_std", F32);
    count(rd_std, "rd_std", N);
    optdev(noise, "noise", F32);
    if (noise.has_value())
        TORCH_CHECK(noise->dim() == 3 && noise->size(0) >= num_iters + (first_draw_proposes ? 0 : 1) && noise->size(1) == B && noise->size(2) == N,
                    "noise must be [>= num_iters + 1 - first_draw_proposes, B, N], got ", noise->sizes());
    dev(obj, "obj", I64);
    count(obj, "obj", B);
    RLS_GUARD(xs);
    ok(rls_maxcut_local_search(G(g), (uint8_t*)p(xs), B, p(ws), wb, ws.size(1), (const float*)p(rd_std), (const float*)p(noise),
                               (uint64_t)seed, env_offset, (int32_t)num_iters, (int32_t)num_spin, first_draw_proposes, (int64_t*)p(obj),
                               compute_obj, cur_stream(xs)), "rls_maxcut_local_search");
}
void maxcut_ls_normals(Tensor out, int64_t seed, int64_t env_offset, int64_t draw) {
    dev(out, "out", F32);
    TORCH_CHECK(out.dim() == 2, "out must be [B, N]");
    RLS_GUARD(out);
    ok(rls_maxcut_ls_normals((float*)p(out), out.size(0), out.size(1), (uint64_t)seed, env_offset, (int32_t)draw, cur_stream(out)),
       "rls_maxcut_ls_normals");
}
static int64_t scratch_of(const OptTensor& scratch) {   // a flat byte buffer on the device, or none
    if (!scratch.has_value()) return 0;
    TORCH_CHECK(scratch->is_cuda() && scratch->scalar_type() == at::kByte && scratch->is_contiguous(), "scratch must be a contiguous uint8 HIP tensor");
    return scratch->numel();
}
void maxcut_ls_threshold(int64_t g, const Tensor& ws, const Tensor& rd_std, int64_t seed, int64_t env_offset, int64_t draw,
                         int64_t num_spin, Tensor thresh, const OptTensor& scratch) {
    const int64_t N = G(g)->num_nodes;
    const int wb = ws_bytes_of(ws, false);
    TORCH_CHECK(ws.dim() == 2 && ws.size(1) >= N, "ws must be [B, >= ", N, "], got ", ws.sizes());
    const int64_t B = ws.size(0);
    dev(rd_std, "rd_std", F32);
    count(rd_std, "rd_std", N);
    dev(thresh, "thresh", F32);
    count(thresh, "thresh", B);
    RLS_GUARD(ws);
    ok(rls_maxcut_ls_threshold(G(g), B, p(ws), wb, ws.size(1), (const float*)p(rd_std), (uint64_t)seed, env_offset, (int32_t)draw, (int32_t)num_spin,
                               (float*)p(thresh), p(scratch), scratch_of(scratch), cur_stream(ws)), "rls_maxcut_ls_threshold");
}
void maxcut_ls_propose(int64_t g, Tensor xs, const Tensor& ws, const Tensor& rd_std, const Tensor& thresh, int64_t seed,
                       int64_t env_offset, int64_t draw, Tensor obj, const OptTensor& scratch) {
    spin_bytes(xs, "xs", false);
    const int64_t B = env_rows(xs, "xs", G(g)), N = G(g)->num_nodes;
    const int wb = ws_bytes_of(ws, false);
    TORCH_CHECK(ws.dim() == 2 && ws.size(0) == B && ws.size(1) >= N, "ws must be [", B, ", >= ", N, "], got ", ws.sizes());
    dev(rd_std, "rd_std", F32);
    count(rd_std, "rd_std", N);
    dev(thresh, "thresh", F32);
    count(thresh, "thresh", B);
    dev(obj, "obj", I64);
    count(obj, "obj", B);
    RLS_GUARD(xs);
    ok(rls_maxcut_ls_propose(G(g), (uint8_t*)p(xs), B, p(ws), wb, ws.size(1), (const float*)p(rd_std), (const float*)p(thresh), (uint64_t)seed,
                             env_offset, (int32_t)draw, (int64_t*)p(obj), p(scratch), scratch_of(scratch), cur_stream(xs)),
       "rls_maxcut_ls_propose");
}
void maxcut_ls_rounds(int64_t g, Tensor xs, const Tensor& ws, const Tensor& rd_std, const Tensor& thresh, int64_t seed,
                      int64_t env_offset, int64_t first_draw, int64_t num_draws, Tensor obj, const OptTensor& scratch) {
    spin_bytes(xs, "xs", false);
    const int64_t B = env_rows(xs, "xs", G(g)), N = G(g)->num_nodes;
    const int wb = ws_bytes_of(ws, false);
    TORCH_CHECK(ws.dim() == 2 && ws.size(0) == B && ws.size(1) >= N, "ws must be [", B, ", >= ", N, "], got ", ws.sizes());
    dev(rd_std, "rd_std", F32);
    count(rd_std, "rd_std", N);
    dev(thresh, "thresh", F32);
    count(thresh, "thresh", B);
    dev(obj, "obj", I64);
    count(obj, "obj", B);
    RLS_GUARD(xs);
    ok(rls_maxcut_ls_rounds(G(g), (uint8_t*)p(xs), B, p(ws), wb, ws.size(1), (const float*)p(rd_std), (const float*)p(thresh), (uint64_t)seed,
                            env_offset, (int32_t)first_draw, (int32_t)num_draws, (int64_t*)p(obj), p(scratch), scratch_of(scratch), cur_stream(xs)),
       "rls_maxcut_ls_rounds");
}
void select_better_rows(Tensor xs0, Tensor vs0, const Tensor& xs1, const Tensor& vs1, bool if_maximize) {
    spin_bytes(xs0, "xs0", false);
    spin_bytes(xs1, "xs1", false);
    TORCH_CHECK(xs0.dim() == 2 && xs1.sizes() == xs0.sizes(), "xs0 and xs1 must be [B, N] of the same shape");
    dev(vs0, "vs0", I64);
    dev(vs1, "vs1", I64);
    count(vs0, "vs0", xs0.size(0));
    count(vs1, "vs1", xs0.size(0));
    RLS_GUARD(xs0);
    ok(rls_select_better_rows((uint8_t*)p(xs0), (int64_t*)p(vs0), (const uint8_t*)p(xs1), (const int64_t*)p(vs1), xs0.size(0), xs0.size(1),
                              if_maximize, cur_stream(xs0)), "rls_select_better_rows");
}
void pick_best_of_repeats(const Tensor& xs, const Tensor& vs, int64_t R, bool if_maximize, Tensor good_xs, Tensor good_vs) {
    spin_bytes(xs, "xs", false);
    spin_bytes(good_xs, "good_xs", false);
    dev(vs, "vs", I64);
    dev(good_vs, "good_vs", I64);
    TORCH_CHECK(R > 0 && xs.dim() == 2 && xs.size(0) % R == 0, "xs must be [R*S, N]");
    const int64_t S = xs.size(0) / R;
    count(vs, "vs", xs.size(0));
    shape2(good_xs, "good_xs", S, xs.size(1));
    count(good_vs, "good_vs", S);
    RLS_GUARD(xs);
    ok(rls_pick_best_of_repeats((const uint8_t*)p(xs), (const int64_t*)p(vs), R, S, xs.size(1), if_maximize, (uint8_t*)p(good_xs),
                                (int64_t*)p(good_vs), cur_stream(xs)), "rls_pick_best_of_repeats");
}
void copy_rows(Tensor xs, const OptTensor& vs, const Tensor& dst, const Tensor& src) {
    spin_bytes(xs, "xs", false);
    TORCH_CHECK(xs.dim() == 2, "xs must be [B, N]");
    optdev(vs, "vs", I64);
    if (vs.has_value()) count(*vs, "vs", xs.size(0));
    dev(dst, "dst", I64);
    dev(src, "src", I64);
    count(src, "src", dst.numel());
    RLS_GUARD(xs);
    ok(rls_copy_rows((uint8_t*)p(xs), (int64_t*)p(vs), xs.size(1), (const int64_t*)p(dst), (const int64_t*)p(src), dst.numel(), cur_stream(xs)),
       "rls_copy_rows");
}
void best_update(const Tensor& xs, const Tensor& vs, bool if_maximize, Tensor best_x, Tensor best_v, Tensor improved, const OptTensor& log_v,
                 int64_t log_index, bool force) {
    spin_bytes(xs, "xs", false);
    dev(vs, "vs");
    const int kind = vs.scalar_type() == I64 ? 0 : (vs.scalar_type() == F32 ? 1 : 2);
    TORCH_CHECK(kind != 2 || vs.scalar_type() == F64, "vs must be int64 / float32 / float64");
    spin_bytes(best_x, "best_x", false);
    dev(best_v, "best_v", F64);
    dev(improved, "improved", U8);
    optdev(log_v, "log_v", F64);
    TORCH_CHECK(xs.dim() == 2 && xs.size(0) == vs.numel(), "xs must be [B, N] with B = vs.numel()");
    count(best_x, "best_x", xs.size(1));
    count(best_v, "best_v", 1);
    count(improved, "improved", 1);
    if (log_v.has_value()) TORCH_CHECK(log_index >= 0 && log_index < log_v->numel(), "log_index outside log_v");
    RLS_GUARD(xs);
    ok(rls_best_update((const uint8_t*)p(xs), p(vs), kind, vs.numel(), best_x.numel(), if_maximize, (uint8_t*)p(best_x), (double*)p(best_v),
                       (uint8_t*)p(improved), (double*)p(log_v), log_index, force, cur_stream(xs)), "rls_best_update");
}
void best_key(const Tensor& vs, int64_t rank_bits, int64_t low_code, int64_t limit, Tensor key, const OptTensor& index, Tensor flag) {
    dev(vs, "vs");
    const auto dt = vs.scalar_type();
    const int kind = dt == I64 ? 0 : (dt == F32 ? 1 : (dt == F64 ? 2 : 3));
    TORCH_CHECK(kind != 3 || dt == I32, "vs must be int64 / int32 / float32 / float64");
    TORCH_CHECK(vs.numel() >= 1, "vs is empty");
    dev(key, "key", I64);
    optdev(index, "index", I64);
    dev(flag, "flag", I32);
    at_least(key, "key", 1);
    at_least(flag, "flag", 1);
    if (index.has_value()) at_least(*index, "index", 1);
    RLS_GUARD(vs);
    ok(rls_best_key(p(vs), kind, vs.numel(), (int32_t)rank_bits, low_code, limit, (int64_t*)p(key), (int64_t*)p(index), (int32_t*)p(flag),
                    cur_stream(vs)), "rls_best_key");
}
void key_unpack(const Tensor& key, int64_t rank_bits, int64_t world, Tensor obj, const OptTensor& owner, int64_t empty_key,
                const OptTensor& flag) {
    dev(key, "key", I64);
    at_least(key, "key", 1);
    dev(obj, "obj");
    TORCH_CHECK(obj.scalar_type() == I64 || obj.scalar_type() == F64, "obj must be int64 or float64");
    at_least(obj, "obj", 1);
    optdev(owner, "owner", I64);
    optdev(flag, "flag", I32);
    if (owner.has_value()) at_least(*owner, "owner", 1);
    if (flag.has_value()) at_least(*flag, "flag", 1);
    same_device(key, obj, "obj");
    RLS_GUARD(key);
    ok(rls_key_unpack((const int64_t*)p(key), (int32_t)rank_bits, world, obj.scalar_type() == F64, p(obj), (int64_t*)p(owner), empty_key,
                      (int32_t*)p(flag), cur_stream(key)), "rls_key_unpack");
}
void winner_message(const OptTensor& xs, const OptTensor& index, const Tensor& key, int64_t rank_bits, int64_t my_low_code, int64_t env_offset,
                    int64_t N, Tensor msg) {
    dev(key, "key", I64);
    at_least(key, "key", 1);
    dev(msg, "msg", at::kByte);
    TORCH_CHECK(msg.is_contiguous() && msg.numel() == 8 + (N + 7) / 8, "msg must hold 8 + ceil(N / 8) bytes");
    optdev(index, "index", I64);
    int64_t B = 0;
    if (xs.has_value()) {
        spin_bytes(*xs, "xs", false);
        TORCH_CHECK((xs->dim() == 2 && xs->size(1) == N) || (xs->dim() == 1 && xs->size(0) == N), "xs must be [B, N] or [N]");
        B = xs->dim() == 2 ? xs->size(0) : 1;
        same_device(key, *xs, "xs");
        TORCH_CHECK(B <= 1 || index.has_value(), "a [B, N] xs needs the row index");
    }
    if (index.has_value()) at_least(*index, "index", 1);
    same_device(key, msg, "msg");
    RLS_GUARD(key);
    ok(rls_winner_message((const uint8_t*)p(xs), B, N, (const int64_t*)p(index), (const int64_t*)p(key), (int32_t)rank_bits, my_low_code,
                          env_offset, (uint8_t*)p(msg), cur_stream(key)), "rls_winner_message");
}
void winner_unpack(const Tensor& msg, int64_t N, const OptTensor& x_out, const OptTensor& index_out) {
    dev(msg, "msg", at::kByte);
    TORCH_CHECK(msg.is_contiguous() && msg.numel() == 8 + (N + 7) / 8, "msg must hold 8 + ceil(N / 8) bytes");
    if (x_out.has_value()) { spin_bytes(*x_out, "x_out", false); TORCH_CHECK(x_out->numel() == N && x_out->is_contiguous(), "x_out must be [N]"); same_device(msg, *x_out, "x_out"); }
    optdev(index_out, "index_out", I64);
    if (index_out.has_value()) at_least(*index_out, "index_out", 1);
    RLS_GUARD(msg);
    ok(rls_winner_unpack((const uint8_t*)p(msg), N, (uint8_t*)p(x_out), (int64_t*)p(index_out), cur_stream(msg)), "rls_winner_unpack");
}
void rand_spins(Tensor x, int64_t seed, int64_t env_offset) {
    spin_bytes(x, "x", false);
    TORCH_CHECK(x.dim() == 2, "x must be [B, N]");
    RLS_GUARD(x);
    ok(rls_rand_spins((uint8_t*)p(x), x.size(0), x.size(1), (uint64_t)seed, env_offset, cur_stream(x)), "rls_rand_spins");
}
void rand_spins_repeats(Tensor x, Tensor repeat_seeds, int64_t env_offset) {
    spin_bytes(x, "x", false);
    dev(repeat_seeds, "repeat_seeds", I64);
    same_device(x, repeat_seeds, "repeat_seeds");
    TORCH_CHECK(x.dim() == 3 && repeat_seeds.dim() == 1 && repeat_seeds.size(0) == x.size(0), "x must be [R, S, N] and repeat_seeds [R]");
    RLS_GUARD(x);
    ok(rls_rand_spins_repeats((uint8_t*)p(x), x.size(0), x.size(1), x.size(2), (const uint64_t*)p(repeat_seeds), env_offset, cur_stream(x)),
       "rls_rand_spins_repeats");
}
void rand_actions(Tensor action, int64_t N, int64_t seed, int64_t step, int64_t env_offset) {
    dev(action, "action", I64);
    RLS_GUARD(action);
    ok(rls_rand_actions((int64_t*)p(action), action.numel(), N, (uint64_t)seed, (uint64_t)step, env_offset, cur_stream(action)), "rls_rand_actions");
}
void rand_perms(Tensor perm, int64_t seed, int64_t env_offset) {
    dev(perm, "perm", I64);
    TORCH_CHECK(perm.dim() == 2, "perm must be [B, N]");
    RLS_GUARD(perm);
    ok(rls_rand_perms((int64_t*)p(perm), perm.size(0), perm.size(1), (uint64_t)seed, env_offset, cur_stream(perm)), "rls_rand_perms");
}

// ------------------------------------------------------------------------------------------------ spin system
inline const rls_spin_env* SE(int64_t handle) {
    TORCH_CHECK(handle != 0, "rlsolver_hip: null spin-env handle");
    return reinterpret_cast<const rls_spin_env*>(handle);
}
inline int state_bytes(const Tensor& state, const rls_spin_env* env) {
    dev(state, "state");
    TORCH_CHECK(state.scalar_type() == F32 || state.scalar_type() == F64, "state must be float32 or float64");
    TORCH_CHECK(state.dim() == 3, "state must be [B, R, N]");
    TORCH_CHECK(env->state == state.data_ptr(), "state is not the buffer the spin-env handle was built on");
    return state.scalar_type() == F64 ? 8 : 4;
}
inline void row_index_ok(const Tensor& row_index) {
    TORCH_CHECK(!row_index.is_cuda() && row_index.scalar_type() == I32 && row_index.numel() == 7 && row_index.is_contiguous(),
                "row_index must be a host int32[7]");
}
void spin_reset(int64_t g, int64_t env, const Tensor& state, const Tensor& row_index, double max_local, int64_t weight_sum) {
    const int sb = state_bytes(state, SE(env));
    row_index_ok(row_index);
    TORCH_CHECK(state.size(2) == G(g)->num_nodes, "state must be [B, R, ", G(g)->num_nodes, "]");
    RLS_GUARD(state);
    ok(rls_spin_reset(G(g), SE(env), sb, state.size(0), (int32_t)state.size(1), (const int32_t*)p(row_index),
                      max_local, weight_sum, cur_stream(state)), "rls_spin_reset");
}
void spin_observation(int64_t env, const Tensor& state, const Tensor& row_index, int64_t step_index, const OptTensor& matrix, bool binary_basis,
                      Tensor out) {
    const int sb = state_bytes(state, SE(env));
    row_index_ok(row_index);
    dev(out, "out", state.scalar_type());
    optdev(matrix, "matrix", state.scalar_type());
    TORCH_CHECK(out.dim() == 3, "state must be [B, R, N], out [B, R (+ N), N]");
    TORCH_CHECK(step_index >= 0 && step_index < SE(env)->table_len, "step_index outside the env's time table");
    const int64_t B = state.size(0), R = state.size(1), N = state.size(2);
    TORCH_CHECK(out.size(0) == B && out.size(2) == N && out.size(1) == R + (matrix.has_value() ? N : 0), "out has the wrong shape");
    bool per_env = false;
    if (matrix.has_value()) {
        per_env = matrix->dim() == 3;
        TORCH_CHECK((matrix->dim() == 2 || (per_env && matrix->size(0) == B)) && matrix->size(-2) == N && matrix->size(-1) == N,
                    "matrix must be [N, N] or [B, N, N]");
    }
    RLS_GUARD(state);
    ok(rls_spin_observation(SE(env), p(matrix), per_env, sb, B, (int32_t)R, N, (const int32_t*)p(row_index), step_index, binary_basis, p(out),
                            cur_stream(state)), "rls_spin_observation");
}
void spin_materialize(int64_t env, Tensor state, const Tensor& row_index, int64_t step_index) {
    const int sb = state_bytes(state, SE(env));
    row_index_ok(row_index);
    TORCH_CHECK(step_index >= 0 && step_index < SE(env)->table_len, "step_index outside the env's time table");
    RLS_GUARD(state);
    ok(rls_spin_materialize(SE(env), sb, state.size(0), state.size(2), (int32_t)state.size(1), (const int32_t*)p(row_index), step_index,
                            cur_stream(state)), "rls_spin_materialize");
}
void rand_couplings(Tensor matrix, int64_t kind, double p_connection, int64_t m_insertion_edges, int64_t edge_type, int64_t seed, int64_t env_offset) {
    dev(matrix, "matrix");
    TORCH_CHECK(matrix.dim() == 3 && matrix.size(1) == matrix.size(2), "matrix must be [B, N, N]");
    TORCH_CHECK(matrix.scalar_type() == F32 || matrix.scalar_type() == F64, "matrix must be float32 or float64");
    RLS_GUARD(matrix);
    ok(rls_rand_couplings(p(matrix), matrix.scalar_type() == F64 ? 8 : 4, matrix.size(0), matrix.size(1), (int32_t)kind, p_connection,
                          (int32_t)m_insertion_edges, (int32_t)edge_type, (uint64_t)seed, env_offset, cur_stream(matrix)), "rls_rand_couplings");
}
void spin_reset_dense(const Tensor& matrix, int64_t env, const Tensor& state, const Tensor& row_index, Tensor max_local, Tensor weight_sum,
                      Tensor flags) {
    const int sb = state_bytes(state, SE(env));
    dev(matrix, "matrix", state.scalar_type());
    dev(max_local, "max_local", state.scalar_type());
    dev(weight_sum, "weight_sum", state.scalar_type());
    dev(flags, "flags", U8);
    row_index_ok(row_index);
    const int64_t B = state.size(0), N = state.size(2);
    shape3(matrix, "matrix", B, N, N);
    TORCH_CHECK(max_local.numel() == B && weight_sum.numel() == B && flags.numel() == B, "max_local / weight_sum / flags must hold B entries");
    RLS_GUARD(state);
    ok(rls_spin_reset_dense(p(matrix), SE(env), sb, B, N, (int32_t)state.size(1), (const int32_t*)p(row_index),
                            p(max_local), p(weight_sum), (uint8_t*)p(flags), cur_stream(state)), "rls_spin_reset_dense");
}
void spin_step_dense(const Tensor& matrix, const Tensor& max_local, int64_t env, const Tensor& state, const Tensor& row_index, const Tensor& action,
                     Tensor reward, const OptTensor& visited_new, double termination_value, int64_t reward_mode,
                     double reward_div, int64_t hist_len, bool use_stag, double stag_punishment, bool use_basin, double basin_reward) {
    const int sb = state_bytes(state, SE(env));
    dev(matrix, "matrix", state.scalar_type());
    dev(max_local, "max_local", state.scalar_type());
    dev(action, "action", I64);
    dev(reward, "reward", state.scalar_type());
    optdev(visited_new, "visited_new", U8);
    row_index_ok(row_index);
    const int64_t B = state.size(0), N = state.size(2);
    shape3(matrix, "matrix", B, N, N);
    TORCH_CHECK(max_local.numel() == B && action.numel() == B && reward.numel() == B, "max_local / action / reward must hold B entries");
    if (visited_new.has_value()) count(*visited_new, "visited_new", B);
    TORCH_CHECK(hist_len >= 0 && (SE(env)->hist == nullptr || hist_len <= SE(env)->hist_cap), "hist_len outside the visited-state ring");
    RLS_GUARD(state);
    ok(rls_spin_step_dense(p(matrix), p(max_local), SE(env), sb, B, N, (int32_t)state.size(1),
                           (const int32_t*)p(row_index), (const int64_t*)p(action), p(reward), (uint8_t*)p(visited_new), termination_value,
                           (int32_t)reward_mode, reward_div, hist_len, use_stag, stag_punishment, use_basin, basin_reward, cur_stream(state)),
       "rls_spin_step_dense");
}
void spin_step(int64_t g, int64_t env, const Tensor& state, const Tensor& row_index, const Tensor& action, Tensor reward,
               const OptTensor& visited_new, double max_local, double termination_value, int64_t reward_mode,
               double reward_div, int64_t hist_len, bool use_stag, double stag_punishment, bool use_basin, double basin_reward) {
    const int sb = state_bytes(state, SE(env));
    dev(action, "action", I64);
    dev(reward, "reward", state.scalar_type());
    optdev(visited_new, "visited_new", U8);
    row_index_ok(row_index);
    const int64_t B = state.size(0);
    TORCH_CHECK(state.size(2) == G(g)->num_nodes, "state must be [B, R, ", G(g)->num_nodes, "]");
    TORCH_CHECK(action.numel() == B && reward.numel() == B, "action / reward must hold B entries");
    if (visited_new.has_value()) count(*visited_new, "visited_new", B);
    TORCH_CHECK(hist_len >= 0 && (SE(env)->hist == nullptr || hist_len <= SE(env)->hist_cap), "hist_len outside the visited-state ring");
    RLS_GUARD(state);
    ok(rls_spin_step(G(g), SE(env), sb, B, (int32_t)state.size(1), (const int32_t*)p(row_index),
                     (const int64_t*)p(action), p(reward), (uint8_t*)p(visited_new), max_local, termination_value, (int32_t)reward_mode,
                     reward_div, hist_len, use_stag, stag_punishment, use_basin, basin_reward, cur_stream(state)), "rls_spin_step");
}

// ------------------------------------------------------------------------------------------------ MCPG
// (N, C) of a chain batch: node-major [N, C], or bit-packed [ceil(C / 64), N] with C given by the caller
inline int64_t chain_nodes(const Tensor& t, int sb, const char* name) {
    TORCH_CHECK(t.dim() == 2, name, " must be 2-D");
    return sb == 0 ? t.size(1) : t.size(0);
}
inline void chain_shape(const Tensor& t, int sb, int64_t N, int64_t C, const char* name) {
    if (sb == 0) shape2(t, name, (C + 63) / 64, N);
    else shape2(t, name, N, C);
}
// (offset, period, skip) of the op schemas -> rls_chain_ids; all zero = NULL (the single-process numbering)
struct ChainIdsArg {
    rls_chain_ids ids;
    bool any;
    ChainIdsArg(int64_t offset, int64_t period, int64_t skip) : ids{offset, period, skip}, any(offset != 0 || period != 0 || skip != 0) {
        TORCH_CHECK(offset >= 0 && period >= 0 && skip >= 0, "chain_offset / chain_period / chain_skip must be >= 0");
    }
    const rls_chain_ids* ptr() const { return any ? &ids : nullptr; }
};
void mcpg_metro_rounds(Tensor samples, const OptTensor& samples_in, int64_t C_in, int64_t C, const Tensor& probs, int64_t T,
                       int64_t t_offset, const OptTensor& index, const OptTensor& u, int64_t seed, const OptTensor& t_limit, bool write_back,
                       const OptTensor& accepts, int64_t chain_offset, int64_t chain_period, int64_t chain_skip, const OptTensor& scratch) {
    const ChainIdsArg cid(chain_offset, chain_period, chain_skip);
    const int sb = chain_bytes(samples, "samples");
    const int64_t N = chain_nodes(samples, sb, "samples");
    chain_shape(samples, sb, N, C, "samples");
    if (samples_in.has_value()) {
        TORCH_CHECK(chain_bytes(*samples_in, "samples_in") == sb, "samples_in must have the layout of samples");
        chain_shape(*samples_in, sb, N, (sb == 0 && C_in > 0) ? C_in : C, "samples_in");
    }
    dev(probs, "probs", F32);
    count(probs, "probs", N);
    optdev(index, "index", I64);
    optdev(u, "u", F32);
    TORCH_CHECK(index.has_value() == u.has_value(), "index and u must be given together");
    if (index.has_value()) {
        TORCH_CHECK(index->dim() == 2 && index->size(0) >= t_offset + T && index->size(1) == C, "index must be [>= t_offset + T, C]");
        TORCH_CHECK(u->dim() == 2 && u->size(0) >= t_offset + T && u->size(1) == C, "u must be [>= t_offset + T, C]");
    }
    optdev(t_limit, "t_limit", I64);
    optdev(accepts, "accepts", I64);
    int64_t accept_rows = 1;      // accepts is [T] or [rows, T]
    if (accepts.has_value()) {
        TORCH_CHECK((accepts->dim() == 1 || accepts->dim() == 2) && accepts->size(-1) == T, "accepts must be [T] or [rows, T]");
        if (accepts->dim() == 2) accept_rows = accepts->size(0);
    }
    int64_t scratch_bytes = 0;
    if (scratch.has_value()) {
        dev(*scratch, "scratch");
        same_device(samples, *scratch, "scratch");
        TORCH_CHECK(scratch->is_contiguous(), "scratch must be contiguous");
        scratch_bytes = (int64_t)scratch->nbytes();
    }
    RLS_GUARD(samples);
    ok(rls_mcpg_metro_rounds(p(samples), p(samples_in), C_in, sb, N, C, (const float*)p(probs), T, t_offset, (const int64_t*)p(index),
                             (const float*)p(u), (uint64_t)seed, (const int64_t*)p(t_limit), write_back, (int64_t*)p(accepts), accept_rows,
                             cid.ptr(), p(scratch), scratch_bytes, cur_stream(samples)),
       "rls_mcpg_metro_rounds");
}
void mcpg_metro_stop(const Tensor& accepts, int64_t target, int64_t first, int64_t next_T, Tensor ctl, const OptTensor& apply_limit) {
    dev(accepts, "accepts", I64);
    TORCH_CHECK(accepts.dim() == 2 && accepts.size(1) >= 1, "accepts must be [rows, T]");
    dev(ctl, "ctl", I64);
    count(ctl, "ctl", 3);
    optdev(apply_limit, "apply_limit", I64);
    if (apply_limit.has_value()) count(*apply_limit, "apply_limit", 1);
    RLS_GUARD(accepts);
    TORCH_CHECK(first >= 0 && first <= 2, "first must be 0, 1 or 2");
    ok(rls_mcpg_metro_stop((const int64_t*)p(accepts), accepts.size(0), accepts.size(1), target, (int32_t)first, next_T, (int64_t*)p(ctl),
                           (int64_t*)p(apply_limit), cur_stream(accepts)), "rls_mcpg_metro_stop");
}
void mcpg_local_search(int64_t g, const Tensor& xs_in, Tensor xs_out, const Tensor& order, const OptTensor& visit_stream, int64_t num_ls,
                       const OptTensor& uniforms, int64_t seed, const OptTensor& edge_weights, int64_t gauge_node, Tensor expected,
                       int64_t chain_offset, int64_t chain_period, int64_t chain_skip) {
    const ChainIdsArg cid(chain_offset, chain_period, chain_skip);
    const int sb = chain_bytes(xs_in, "xs_in");
    TORCH_CHECK(sb != 0, "rls_mcpg_local_search takes node-major chains");
    const int64_t N = G(g)->num_nodes;
    TORCH_CHECK(xs_in.dim() == 2 && xs_in.size(0) == N, "xs_in must be [", N, ", C]");
    const int64_t C = xs_in.size(1);
    dev(xs_out, "xs_out", F32);
    shape2(xs_out, "xs_out", N, C);
    dev(order, "order", I32);
    count(order, "order", N);
    optdev(visit_stream, "visit_stream", I32);
    optdev(uniforms, "uniforms", F32);
    if (uniforms.has_value()) shape3(*uniforms, "uniforms", num_ls, N, C);
    optdev(edge_weights, "edge_weights", I32);
    if (edge_weights.has_value()) count(*edge_weights, "edge_weights", G(g)->num_stored_edges);
    TORCH_CHECK(gauge_node >= -1 && gauge_node < N, "gauge_node outside [-1, N)");
    dev(expected, "expected", F32);
    count(expected, "expected", C);
    RLS_GUARD(xs_in);
    ok(rls_mcpg_local_search(G(g), p(xs_in), sb, (float*)p(xs_out), C, (const int32_t*)p(order), (const int32_t*)p(visit_stream),
                             visit_stream.has_value() ? visit_stream->numel() : 0, num_ls, (const float*)p(uniforms), (uint64_t)seed,
                             (const int32_t*)p(edge_weights), gauge_node, (float*)p(expected), cid.ptr(), cur_stream(xs_in)), "rls_mcpg_local_search");
}
void mcpg_local_search_levels(int64_t g, const Tensor& xs_in, int64_t C_in, Tensor xs_out, int64_t C, const Tensor& lv_ptr,
                              const Tensor& lv_data, int64_t num_ls, const OptTensor& coins, int64_t seed, Tensor expected,
                              int64_t chain_offset, int64_t chain_period, int64_t chain_skip) {
    const ChainIdsArg cid(chain_offset, chain_period, chain_skip);
    const int sb = chain_bytes(xs_in, "xs_in"), osb = chain_bytes(xs_out, "xs_out");
    const int64_t N = G(g)->num_nodes;
    chain_shape(xs_in, sb, N, C_in > 0 ? C_in : C, "xs_in");
    TORCH_CHECK(osb == 0 || osb == 4, "xs_out must be float32 node-major or bit-packed");
    chain_shape(xs_out, osb, N, C, "xs_out");
    dev(lv_ptr, "lv_ptr", I32);
    dev(lv_data, "lv_data", I32);
    TORCH_CHECK(lv_ptr.numel() >= 1, "lv_ptr must hold groups + 1 offsets");
    optdev(coins, "coins", I64);
    if (coins.has_value()) shape2(*coins, "coins", num_ls * N, (C + 63) / 64);
    dev(expected, "expected", F32);
    count(expected, "expected", C);
    RLS_GUARD(xs_in);
    ok(rls_mcpg_local_search_levels(G(g), p(xs_in), sb, C_in, p(xs_out), osb, C, (const int32_t*)p(lv_ptr), (const int32_t*)p(lv_data),
                                    lv_ptr.numel() - 1, num_ls, (const uint64_t*)p(coins), (uint64_t)seed, (float*)p(expected), cid.ptr(), cur_stream(xs_in)),
       "rls_mcpg_local_search_levels");
}
void mcpg_pick_best(const Tensor& expected, const Tensor& xs, int64_t N, int64_t total_mcmc_num, int64_t repeat_times, int64_t num_edges,
                    Tensor best_index, Tensor vs_good, Tensor xs_good) {
    dev(expected, "expected", F32);
    const int sb = chain_bytes(xs, "xs");
    TORCH_CHECK(chain_bytes(xs_good, "xs_good") == sb, "xs_good must have the layout of xs");
    TORCH_CHECK(total_mcmc_num > 0 && repeat_times > 0, "total_mcmc_num and repeat_times must be positive");
    chain_shape(xs, sb, N, total_mcmc_num * repeat_times, "xs");
    chain_shape(xs_good, sb, N, total_mcmc_num, "xs_good");
    count(expected, "expected", total_mcmc_num * repeat_times);
    dev(best_index, "best_index", I64);
    dev(vs_good, "vs_good", F32);
    count(best_index, "best_index", total_mcmc_num);
    count(vs_good, "vs_good", total_mcmc_num);
    RLS_GUARD(xs);
    ok(rls_mcpg_pick_best((const float*)p(expected), p(xs), sb, N, total_mcmc_num, repeat_times, num_edges, (int64_t*)p(best_index),
                          (float*)p(vs_good), p(xs_good), cur_stream(xs)), "rls_mcpg_pick_best");
}
void mcpg_merge_best(const Tensor& temp_max, Tensor temp_info, Tensor now_max_res, Tensor now_info, int64_t total_mcmc_num, Tensor mask_scratch,
                     const OptTensor& best_value, const OptTensor& best_index, bool replace_worst) {
    dev(temp_max, "temp_max", F32);
    dev(temp_info, "temp_info", I64);
    dev(now_max_res, "now_max_res", F32);
    dev(now_info, "now_info", I64);
    dev(mask_scratch, "mask_scratch", I64);
    optdev(best_value, "best_value", F32);
    optdev(best_index, "best_index", I64);
    const int64_t M = total_mcmc_num, tiles = (M + 63) / 64;
    TORCH_CHECK(M > 0 && temp_info.dim() == 2 && temp_info.size(0) == tiles, "temp_info must be [ceil(M / 64), N]");
    TORCH_CHECK(now_info.sizes() == temp_info.sizes(), "now_info must have the shape of temp_info");
    count(temp_max, "temp_max", M);
    count(now_max_res, "now_max_res", M);
    at_least(mask_scratch, "mask_scratch", tiles);
    if (best_value.has_value()) at_least(*best_value, "best_value", replace_worst ? 1 : 2);
    if (best_index.has_value()) at_least(*best_index, "best_index", replace_worst ? 1 : 2);
    TORCH_CHECK(replace_worst || (best_value.has_value() && best_index.has_value()), "replace_worst = False reports {max, min} in best_value / best_index [2]");
    RLS_GUARD(temp_info);
    ok(rls_mcpg_merge_best((const float*)p(temp_max), (uint64_t*)p(temp_info), (float*)p(now_max_res), (uint64_t*)p(now_info), temp_info.size(1),
                           total_mcmc_num, (uint64_t*)p(mask_scratch), (float*)p(best_value), (int64_t*)p(best_index), replace_worst ? 1 : 0,
                           cur_stream(temp_info)),
       "rls_mcpg_merge_best");
}
void mcpg_value_bit_sums(const Tensor& samples, int64_t C, const Tensor& value, Tensor A) {
    dev(samples, "samples", I64);
    dev(value, "value", F32);
    dev(A, "A", F32);
    TORCH_CHECK(samples.dim() == 2 && samples.size(0) == (C + 63) / 64, "samples must be [ceil(C / 64), N]");
    count(value, "value", C);
    count(A, "A", samples.size(1));
    RLS_GUARD(samples);
    ok(rls_mcpg_value_bit_sums((const uint64_t*)p(samples), samples.size(1), C, (const float*)p(value), (float*)p(A), cur_stream(samples)),
       "rls_mcpg_value_bit_sums");
}
void mcpg_pack_chains(const Tensor& xs, Tensor packed) {
    const int sb = chain_bytes(xs, "xs");
    TORCH_CHECK(sb != 0 && xs.dim() == 2, "xs must be node-major [N, C]");
    dev(packed, "packed", I64);
    shape2(packed, "packed", (xs.size(1) + 63) / 64, xs.size(0));
    RLS_GUARD(xs);
    ok(rls_mcpg_pack_chains(p(xs), sb, xs.size(0), xs.size(1), (uint64_t*)p(packed), cur_stream(xs)), "rls_mcpg_pack_chains");
}
void mcpg_unpack_chains(const Tensor& packed, int64_t C, Tensor xs) {
    dev(packed, "packed", I64);
    dev(xs, "xs", F32);
    TORCH_CHECK(packed.dim() == 2 && packed.size(0) == (C + 63) / 64, "packed must be [ceil(C / 64), N]");
    shape2(xs, "xs", packed.size(1), C);
    RLS_GUARD(packed);
    ok(rls_mcpg_unpack_chains((const uint64_t*)p(packed), packed.size(1), C, (float*)p(xs), cur_stream(packed)), "rls_mcpg_unpack_chains");
}
void qubo_local_search_value(const Tensor& Q, const Tensor& xs_in, Tensor xs_out, int64_t num_ls, bool binary, Tensor value) {
    dev(Q, "Q", F32);
    dev(xs_in, "xs_in", F32);
    dev(xs_out, "xs_out", F32);
    dev(value, "value", F32);
    TORCH_CHECK(Q.dim() == 2 && Q.size(0) == Q.size(1), "Q must be [n, n]");
    TORCH_CHECK(xs_in.dim() == 2 && xs_in.size(0) == Q.size(0), "xs_in must be [n, C]");
    TORCH_CHECK(xs_out.sizes() == xs_in.sizes(), "xs_out must have the shape of xs_in");
    count(value, "value", xs_in.size(1));
    RLS_GUARD(Q);
    ok(rls_qubo_local_search_value((const float*)p(Q), Q.size(0), (const float*)p(xs_in), (float*)p(xs_out), xs_in.size(1), num_ls, binary,
                                   (float*)p(value), cur_stream(Q)), "rls_qubo_local_search_value");
}

void qubo_sparse_local_search_value(const Tensor& rowptr, const Tensor& col, const Tensor& val, const OptTensor& lv_ptr,
                                    const OptTensor& lv_rows, const Tensor& xs_in, Tensor xs_out, int64_t num_ls, bool binary, Tensor value) {
    dev(rowptr, "rowptr", I32);
    optdev(lv_ptr, "lv_ptr", I32);
    optdev(lv_rows, "lv_rows", I32);
    TORCH_CHECK(lv_ptr.has_value() == lv_rows.has_value(), "lv_ptr and lv_rows come together");
    if (lv_ptr.has_value()) {
        TORCH_CHECK(lv_ptr->numel() >= 2 && lv_rows->numel() == rowptr.numel() - 1, "lv_ptr must be [levels + 1], lv_rows [n]");
        same_device(xs_in, *lv_ptr, "lv_ptr");
        same_device(xs_in, *lv_rows, "lv_rows");
    }
    dev(col, "col", I32);
    dev(val, "val", F32);
    dev(xs_in, "xs_in", F32);
    dev(xs_out, "xs_out", F32);
    dev(value, "value", F32);
    TORCH_CHECK(rowptr.numel() >= 1 && xs_in.dim() == 2 && xs_in.size(0) == rowptr.numel() - 1, "xs_in must be [n, C] with n = rowptr.numel() - 1");
    TORCH_CHECK(col.numel() == val.numel(), "col and val must hold nnz entries each");
    TORCH_CHECK(xs_out.sizes() == xs_in.sizes(), "xs_out must have the shape of xs_in");
    count(value, "value", xs_in.size(1));
    RLS_GUARD(xs_in);
    ok(rls_qubo_sparse_local_search_value((const int32_t*)p(rowptr), (const int32_t*)p(col), (const float*)p(val), rowptr.numel() - 1,
                                          (const int32_t*)p(lv_ptr), (const int32_t*)p(lv_rows),
                                          lv_ptr.has_value() ? (int32_t)(lv_ptr->numel() - 1) : 0, (const float*)p(xs_in), (float*)p(xs_out), xs_in.size(1), num_ls, binary, (float*)p(value),
                                          cur_stream(xs_in)), "rls_qubo_sparse_local_search_value");
}

// ------------------------------------------------------------------------------------------------ TSP / ISCO
inline void perm_and_dist(const Tensor& dist, const Tensor& perm, at::ScalarType dist_dt) {
    dev(dist, "dist", dist_dt);
    dev(perm, "perm", I64);
    TORCH_CHECK(perm.dim() == 2, "perm must be [B, N]");
    shape2(dist, "dist", perm.size(1), perm.size(1));
}
void tsp_tour_length(const Tensor& dist, const Tensor& perm, Tensor length) {
    perm_and_dist(dist, perm, F32);
    dev(length, "length", F32);
    count(length, "length", perm.size(0));
    RLS_GUARD(perm);
    ok(rls_tsp_tour_length((const float*)p(dist), perm.size(1), (const int64_t*)p(perm), perm.size(0), (float*)p(length), cur_stream(perm)),
       "rls_tsp_tour_length");
}
void tsp_swap_delta_all(const Tensor& dist, const Tensor& perm, const OptTensor& selected, const OptTensor& nearest, const OptTensor& random,
                        const OptTensor& tables8, double near_threshold, int64_t seed, int64_t env_offset, const OptTensor& selected_out, double temperature,
                        Tensor logratio, Tensor indices, Tensor ban) {
    perm_and_dist(dist, perm, F32);
    optdev(selected, "selected", I64);
    optdev(nearest, "nearest", I32);
    optdev(random, "random", I32);
    optdev(selected_out, "selected_out", I64);
    dev(logratio, "logratio", F32);
    dev(indices, "indices", I64);
    spin_bytes(ban, "ban", false);
    const int64_t N = perm.size(1);
    TORCH_CHECK(logratio.sizes() == perm.sizes() && indices.sizes() == perm.sizes() && ban.sizes() == perm.sizes(),
                "logratio / indices / ban must have the shape of perm");
    int32_t K = 0, stride = 0;
    if (selected.has_value()) {
        TORCH_CHECK(selected->sizes() == perm.sizes(), "selected must have the shape of perm");
        TORCH_CHECK(!selected_out.has_value(), "selected_out records the in-kernel draw: it needs selected = None");
    } else {
        TORCH_CHECK(nearest.has_value() && random.has_value(), "selected = None draws the partners in the kernel: nearest / random must be given");
        TORCH_CHECK(nearest->dim() == 2 && nearest->size(0) == N && random->dim() == 2 && random->size(0) == N, "nearest / random must be [N, *]");
        same_device(perm, *nearest, "nearest");
        same_device(perm, *random, "random");
        K = (int32_t)nearest->size(1);
        stride = (int32_t)random->size(1);
        if (selected_out.has_value()) TORCH_CHECK(selected_out->sizes() == perm.sizes(), "selected_out must have the shape of perm");
        if (tables8.has_value()) {
            dev(*tables8, "tables8", at::kByte);
            same_device(perm, *tables8, "tables8");
            TORCH_CHECK(tables8->is_contiguous() && tables8->numel() == rls_tsp_tables8_bytes(N, K) && tables8->numel() > 0,
                        "tables8 must hold rls_tsp_tables8_bytes(N, K) bytes");
        }
    }
    RLS_GUARD(perm);
    ok(rls_tsp_swap_delta_all((const float*)p(dist), N, (const int64_t*)p(perm), perm.size(0), (const int64_t*)p(selected),
                              (const int32_t*)p(nearest), K, (const int32_t*)p(random), stride, (const uint8_t*)p(tables8),
                              (float)near_threshold, (uint64_t)seed,
                              env_offset, (int64_t*)p(selected_out), (float)temperature, (float*)p(logratio), (int64_t*)p(indices),
                              (uint8_t*)p(ban), cur_stream(perm)),
       "rls_tsp_swap_delta_all");
}
void tsp_apply_swap(Tensor perm, const Tensor& pos, const Tensor& indices) {
    dev(perm, "perm", I64);
    dev(pos, "pos", I64);
    dev(indices, "indices", I64);
    TORCH_CHECK(perm.dim() == 2 && indices.sizes() == perm.sizes(), "perm and indices must be [B, N]");
    count(pos, "pos", perm.size(0));
    RLS_GUARD(perm);
    ok(rls_tsp_apply_swap((int64_t*)p(perm), perm.size(0), perm.size(1), (const int64_t*)p(pos), (const int64_t*)p(indices), cur_stream(perm)),
       "rls_tsp_apply_swap");
}
void tsp_2opt_best(const Tensor& dist, const Tensor& perm, const OptTensor& cur_length, Tensor best_i, Tensor best_j, Tensor best_value) {
    perm_and_dist(dist, perm, F64);
    optdev(cur_length, "cur_length", F64);
    dev(best_i, "best_i", I64);
    dev(best_j, "best_j", I64);
    dev(best_value, "best_value", F64);
    const int64_t slices = perm.size(0) > 0 ? best_i.numel() / perm.size(0) : 1;
    TORCH_CHECK(slices >= 1 && best_i.numel() == slices * perm.size(0) && best_j.numel() == best_i.numel() && best_value.numel() == best_i.numel(),
                "outputs must hold slices * B entries each");
    if (cur_length.has_value()) TORCH_CHECK(cur_length->numel() == perm.size(0), "cur_length must hold B entries");
    RLS_GUARD(perm);
    ok(rls_tsp_2opt_best((const double*)p(dist), perm.size(1), (const int64_t*)p(perm), perm.size(0), (const double*)p(cur_length), (int32_t)slices,
                         (int64_t*)p(best_i), (int64_t*)p(best_j), (double*)p(best_value), cur_stream(perm)), "rls_tsp_2opt_best");
}
void tsp_2opt_delta(const Tensor& dist, const Tensor& perm, const Tensor& i, const Tensor& j, Tensor delta) {
    perm_and_dist(dist, perm, F32);
    dev(i, "i", I64);
    dev(j, "j", I64);
    dev(delta, "delta", F32);
    count(i, "i", perm.size(0));
    count(j, "j", perm.size(0));
    count(delta, "delta", perm.size(0));
    RLS_GUARD(perm);
    ok(rls_tsp_2opt_delta((const float*)p(dist), perm.size(1), (const int64_t*)p(perm), perm.size(0), (const int64_t*)p(i), (const int64_t*)p(j),
                          (float*)p(delta), cur_stream(perm)), "rls_tsp_2opt_delta");
}
void isco_maxcut_step(int64_t g, const Tensor& x, Tensor y_out, const Tensor& path_length, double temperature, const OptTensor& u_gumbel,
                      const OptTensor& u_accept, int64_t seed, int64_t env_offset, const OptTensor& energy_out, const OptTensor& acc_out,
                      const OptTensor& terms_out, const OptTensor& mask_out, const OptTensor& scratch) {
    dev(x, "x", F32);
    dev(y_out, "y_out", F32);
    dev(path_length, "path_length", I64);
    optdev(u_gumbel, "u_gumbel", F32);
    optdev(u_accept, "u_accept", F32);
    optdev(energy_out, "energy_out", F32);
    optdev(acc_out, "acc_out", F32);
    optdev(terms_out, "terms_out", F32);
    const int64_t B = env_rows(x, "x", G(g)), N = G(g)->num_nodes;
    TORCH_CHECK(y_out.sizes() == x.sizes(), "y_out must have the shape of x");
    count(path_length, "path_length", B);
    TORCH_CHECK(u_gumbel.has_value() == u_accept.has_value(), "u_gumbel and u_accept must be given together");
    if (u_gumbel.has_value()) { shape2(*u_gumbel, "u_gumbel", B, N); count(*u_accept, "u_accept", B); }
    if (energy_out.has_value()) count(*energy_out, "energy_out", B);
    if (acc_out.has_value()) count(*acc_out, "acc_out", B);
    if (terms_out.has_value()) shape2(*terms_out, "terms_out", B, 5);
    if (mask_out.has_value()) { spin_bytes(*mask_out, "mask_out", false); shape2(*mask_out, "mask_out", B, N); }
    if (scratch.has_value()) {      // the kernel's f32 rows past ~15 900 nodes: a device pointer it writes through
        dev(*scratch, "scratch");
        same_device(x, *scratch, "scratch");
        TORCH_CHECK(scratch->is_contiguous(), "scratch must be contiguous");
    }
    RLS_GUARD(x);
    ok(rls_isco_maxcut_step(G(g), (const float*)p(x), (float*)p(y_out), B, (const int64_t*)p(path_length), (float)temperature,
                            (const float*)p(u_gumbel), (const float*)p(u_accept), (uint64_t)seed, env_offset, (float*)p(energy_out),
                            (float*)p(acc_out), (float*)p(terms_out), (uint8_t*)p(mask_out), p(scratch),
                            scratch.has_value() ? (int64_t)scratch->nbytes() : 0, cur_stream(x)), "rls_isco_maxcut_step");
}
void isco_tsp_step(const Tensor& dist, const Tensor& nearest, double near_threshold, const Tensor& random, const Tensor& perm_in, Tensor perm_out,
                   int64_t path_length, double temperature, const OptTensor& u_partner, const OptTensor& r_near, const OptTensor& r_rand,
                   const OptTensor& u_gumbel, const OptTensor& u_accept, int64_t seed, int64_t env_offset, const OptTensor& log_acc_out,
                   const OptTensor& acc_out, const OptTensor& cur_out) {
    perm_and_dist(dist, perm_in, F32);
    dev(nearest, "nearest", I32);
    dev(random, "random", I32);
    dev(perm_out, "perm_out", I64);
    optdev(u_partner, "u_partner", F32);
    optdev(r_near, "r_near", I64);
    optdev(r_rand, "r_rand", I64);
    optdev(u_gumbel, "u_gumbel", F32);
    optdev(u_accept, "u_accept", F32);
    optdev(log_acc_out, "log_acc_out", F32);
    optdev(acc_out, "acc_out", F32);
    optdev(cur_out, "cur_out", I64);
    const int64_t B = perm_in.size(0), N = perm_in.size(1);
    TORCH_CHECK(perm_out.sizes() == perm_in.sizes() && perm_out.data_ptr() != perm_in.data_ptr(), "perm_out must be a second [B, N] buffer");
    TORCH_CHECK(nearest.dim() == 2 && nearest.size(0) == N && random.dim() == 2 && random.size(0) == N, "nearest / random must be [N, *]");
    const bool draws = u_partner.has_value();
    TORCH_CHECK(r_near.has_value() == draws && r_rand.has_value() == draws && u_gumbel.has_value() == draws && u_accept.has_value() == draws,
                "the test draws come all or none");
    if (draws) {
        shape3(*u_partner, "u_partner", path_length, B, N);
        shape3(*r_near, "r_near", path_length, B, N);
        shape3(*r_rand, "r_rand", path_length, B, N);
        shape3(*u_gumbel, "u_gumbel", path_length, B, N);
        count(*u_accept, "u_accept", B);
    }
    if (log_acc_out.has_value()) count(*log_acc_out, "log_acc_out", B);
    if (acc_out.has_value()) count(*acc_out, "acc_out", B);
    if (cur_out.has_value()) shape2(*cur_out, "cur_out", B, N);
    RLS_GUARD(perm_in);
    ok(rls_isco_tsp_step((const float*)p(dist), N, (const int32_t*)p(nearest), (int32_t)nearest.size(1), (float)near_threshold,
                         (const int32_t*)p(random), (int32_t)random.size(1), (const int64_t*)p(perm_in), (int64_t*)p(perm_out), B,
                         (int32_t)path_length, (float)temperature, (const float*)p(u_partner), (const int64_t*)p(r_near), (const int64_t*)p(r_rand),
                         (const float*)p(u_gumbel), (const float*)p(u_accept), (uint64_t)seed, env_offset, (float*)p(log_acc_out),
                         (float*)p(acc_out), (int64_t*)p(cur_out), cur_stream(perm_in)), "rls_isco_tsp_step");
}

}  // namespace

// name -> C-ABI function: the table tests/test_abi.py compares with the header's device entry points
TORCH_LIBRARY(rlsolver_hip, m) {
    m.def("maxcut_obj(int graph, Tensor xs, Tensor(a!) obj) -> ()");
    m.def("maxcut_edge_cut_mask(int graph, Tensor xs, Tensor(a!) mask) -> ()");
    m.def("maxcut_node_cutdeg(int graph, Tensor xs, Tensor(a!) out) -> ()");
    m.def("maxcut_delta_all(int graph, Tensor xs, Tensor(a!) out) -> ()");
    m.def("maxcut_step(int graph, Tensor x_in, Tensor(a!) x_out, Tensor action, Tensor(b!) obj, Tensor(c!) reward, Tensor(d!)? cur, "
          "Tensor(e!)? done, float done_value) -> ()");
    m.def("maxcut_greedy_sweep(int graph, Tensor(a!) xs, Tensor(b!) obj) -> ()");
    m.def("maxcut_propose_accept(int graph, Tensor(a!) xs, Tensor mask, Tensor(b!) obj) -> ()");
    m.def("maxcut_ls_weights(int graph, Tensor xs, int mult, Tensor(a!) ws, Tensor(b!)? ws_minmax) -> ()");
    m.def("maxcut_local_search(int graph, Tensor(a!) xs, Tensor ws, Tensor rd_std, Tensor? noise, int seed, int env_offset, int num_iters, "
          "int num_spin, bool first_draw_proposes, Tensor(b!) obj, bool compute_obj) -> ()");
    m.def("maxcut_ls_normals(Tensor(a!) out, int seed, int env_offset, int draw) -> ()");
    m.def("maxcut_ls_threshold(int graph, Tensor ws, Tensor rd_std, int seed, int env_offset, int draw, int num_spin, Tensor(a!) thresh, Tensor(b!)? scratch) -> ()");
    m.def("maxcut_ls_propose(int graph, Tensor(a!) xs, Tensor ws, Tensor rd_std, Tensor thresh, int seed, int env_offset, int draw, "
          "Tensor(b!) obj, Tensor(c!)? scratch) -> ()");
    m.def("maxcut_ls_rounds(int graph, Tensor(a!) xs, Tensor ws, Tensor rd_std, Tensor thresh, int seed, int env_offset, int first_draw, "
          "int num_draws, Tensor(b!) obj, Tensor(c!)? scratch) -> ()");
    m.def("select_better_rows(Tensor(a!) xs0, Tensor(b!) vs0, Tensor xs1, Tensor vs1, bool if_maximize) -> ()");
    m.def("pick_best_of_repeats(Tensor xs, Tensor vs, int R, bool if_maximize, Tensor(a!) good_xs, Tensor(b!) good_vs) -> ()");
    m.def("copy_rows(Tensor(a!) xs, Tensor(b!)? vs, Tensor dst, Tensor src) -> ()");
    m.def("best_update(Tensor xs, Tensor vs, bool if_maximize, Tensor(a!) best_x, Tensor(b!) best_v, Tensor(c!) improved, Tensor(d!)? log_v, "
          "int log_index, bool force) -> ()");
    m.def("best_key(Tensor vs, int rank_bits, int low_code, int limit, Tensor(a!) key, Tensor(b!)? index, Tensor(c!) flag) -> ()");
    m.def("key_unpack(Tensor key, int rank_bits, int world, Tensor(a!) obj, Tensor(b!)? owner, int empty_key, Tensor(c!)? flag) -> ()");
    m.def("winner_message(Tensor? xs, Tensor? index, Tensor key, int rank_bits, int my_low_code, int env_offset, int N, Tensor(a!) msg) -> ()");
    m.def("winner_unpack(Tensor msg, int N, Tensor(a!)? x_out, Tensor(b!)? index_out) -> ()");
    m.def("rand_spins(Tensor(a!) x, int seed, int env_offset) -> ()");
    m.def("rand_spins_repeats(Tensor(a!) x, Tensor repeat_seeds, int env_offset) -> ()");
    m.def("rand_actions(Tensor(a!) action, int N, int seed, int step, int env_offset) -> ()");
    m.def("rand_perms(Tensor(a!) perm, int seed, int env_offset) -> ()");
    m.def("spin_reset(int graph, int env, Tensor(a!) state, Tensor row_index, float max_local, int weight_sum) -> ()");
    m.def("spin_observation(int env, Tensor state, Tensor row_index, int step_index, Tensor? matrix, bool binary_basis, Tensor(a!) out) -> ()");
    m.def("spin_materialize(int env, Tensor(a!) state, Tensor row_index, int step_index) -> ()");
    m.def("rand_couplings(Tensor(a!) matrix, int kind, float p_connection, int m_insertion_edges, int edge_type, int seed, int env_offset) -> ()");
    m.def("spin_reset_dense(Tensor matrix, int env, Tensor(a!) state, Tensor row_index, Tensor(b!) max_local, Tensor(c!) weight_sum, Tensor(d!) flags) -> ()");
    m.def("spin_step_dense(Tensor matrix, Tensor max_local, int env, Tensor(a!) state, Tensor row_index, Tensor action, Tensor(b!) reward, "
          "Tensor(c!)? visited_new, float termination_value, int reward_mode, float reward_div, int hist_len, bool use_stag, "
          "float stag_punishment, bool use_basin, float basin_reward) -> ()");
    m.def("spin_step(int graph, int env, Tensor(a!) state, Tensor row_index, Tensor action, Tensor(b!) reward, Tensor(c!)? visited_new, "
          "float max_local, float termination_value, int reward_mode, float reward_div, int hist_len, bool use_stag, "
          "float stag_punishment, bool use_basin, float basin_reward) -> ()");
    m.def("mcpg_metro_rounds(Tensor(a!) samples, Tensor? samples_in, int C_in, int C, Tensor probs, int T, int t_offset, Tensor? index, "
          "Tensor? u, int seed, Tensor? t_limit, bool write_back, Tensor(b!)? accepts, int chain_offset=0, int chain_period=0, int chain_skip=0, "
          "Tensor(c!)? scratch=None) -> ()");
    m.def("mcpg_metro_stop(Tensor accepts, int target, int first, int next_T, Tensor(a!) ctl, Tensor(b!)? apply_limit) -> ()");
    m.def("mcpg_local_search(int graph, Tensor xs_in, Tensor(a!) xs_out, Tensor order, Tensor? visit_stream, int num_ls, Tensor? uniforms, "
          "int seed, Tensor? edge_weights, int gauge_node, Tensor(b!) expected, int chain_offset=0, int chain_period=0, int chain_skip=0) -> ()");
    m.def("mcpg_local_search_levels(int graph, Tensor xs_in, int C_in, Tensor(a!) xs_out, int C, Tensor lv_ptr, Tensor lv_data, int num_ls, "
          "Tensor? coins, int seed, Tensor(b!) expected, int chain_offset=0, int chain_period=0, int chain_skip=0) -> ()");
    m.def("mcpg_pick_best(Tensor expected, Tensor xs, int N, int total_mcmc_num, int repeat_times, int num_edges, Tensor(a!) best_index, "
          "Tensor(b!) vs_good, Tensor(c!) xs_good) -> ()");
    m.def("mcpg_merge_best(Tensor temp_max, Tensor(a!) temp_info, Tensor(b!) now_max_res, Tensor(c!) now_info, int total_mcmc_num, "
          "Tensor(d!) mask_scratch, Tensor(e!)? best_value, Tensor(f!)? best_index, bool replace_worst=True) -> ()");
    m.def("mcpg_value_bit_sums(Tensor samples, int C, Tensor value, Tensor(a!) A) -> ()");
    m.def("mcpg_pack_chains(Tensor xs, Tensor(a!) packed) -> ()");
    m.def("mcpg_unpack_chains(Tensor packed, int C, Tensor(a!) xs) -> ()");
    m.def("qubo_local_search_value(Tensor Q, Tensor xs_in, Tensor(a!) xs_out, int num_ls, bool binary, Tensor(b!) value) -> ()");
    m.def("qubo_sparse_local_search_value(Tensor rowptr, Tensor col, Tensor val, Tensor? lv_ptr, Tensor? lv_rows, Tensor xs_in, Tensor(a!) xs_out, int num_ls, bool binary, "
          "Tensor(b!) value) -> ()");
    m.def("tsp_tour_length(Tensor dist, Tensor perm, Tensor(a!) length) -> ()");
    m.def("tsp_swap_delta_all(Tensor dist, Tensor perm, Tensor? selected, Tensor? nearest, Tensor? random, Tensor? tables8, float near_threshold, int seed, "
          "int env_offset, Tensor(d!)? selected_out, float temperature, Tensor(a!) logratio, Tensor(b!) indices, Tensor(c!) ban) -> ()");
    m.def("tsp_apply_swap(Tensor(a!) perm, Tensor pos, Tensor indices) -> ()");
    m.def("tsp_2opt_delta(Tensor dist, Tensor perm, Tensor i, Tensor j, Tensor(a!) delta) -> ()");
    m.def("tsp_2opt_best(Tensor dist, Tensor perm, Tensor? cur_length, Tensor(a!) best_i, Tensor(b!) best_j, Tensor(c!) best_value) -> ()");
    m.def("isco_maxcut_step(int graph, Tensor x, Tensor(a!) y_out, Tensor path_length, float temperature, Tensor? u_gumbel, Tensor? u_accept, "
          "int seed, int env_offset, Tensor(b!)? energy_out, Tensor(c!)? acc_out, Tensor(d!)? terms_out, Tensor(e!)? mask_out, Tensor(f!)? scratch=None) -> ()");
    m.def("isco_tsp_step(Tensor dist, Tensor nearest, float near_threshold, Tensor random, Tensor perm_in, Tensor(a!) perm_out, int path_length, "
          "float temperature, Tensor? u_partner, Tensor? r_near, Tensor? r_rand, Tensor? u_gumbel, Tensor? u_accept, int seed, int env_offset, "
          "Tensor(b!)? log_acc_out, Tensor(c!)? acc_out, Tensor(d!)? cur_out) -> ()");
}

TORCH_LIBRARY_IMPL(rlsolver_hip, CUDA, m) {   // "CUDA" is the HIP dispatch key on PyTorch-ROCm
    m.impl("maxcut_obj", &maxcut_obj);
    m.impl("maxcut_edge_cut_mask", &maxcut_edge_cut_mask);
    m.impl("maxcut_node_cutdeg", &maxcut_node_cutdeg);
    m.impl("maxcut_delta_all", &maxcut_delta_all);
    m.impl("maxcut_step", &maxcut_step);
    m.impl("maxcut_greedy_sweep", &maxcut_greedy_sweep);
    m.impl("maxcut_propose_accept", &maxcut_propose_accept);
    m.impl("maxcut_ls_weights", &maxcut_ls_weights);
    m.impl("maxcut_local_search", &maxcut_local_search);
    m.impl("maxcut_ls_normals", &maxcut_ls_normals);
    m.impl("maxcut_ls_threshold", &maxcut_ls_threshold);
    m.impl("maxcut_ls_propose", &maxcut_ls_propose);
    m.impl("maxcut_ls_rounds", &maxcut_ls_rounds);
    m.impl("select_better_rows", &select_better_rows);
    m.impl("pick_best_of_repeats", &pick_best_of_repeats);
    m.impl("copy_rows", &copy_rows);
    m.impl("best_update", &best_update);
    m.impl("best_key", &best_key);
    m.impl("key_unpack", &key_unpack);
    m.impl("winner_message", &winner_message);
    m.impl("winner_unpack", &winner_unpack);
    m.impl("rand_spins", &rand_spins);
    m.impl("rand_spins_repeats", &rand_spins_repeats);
    m.impl("rand_actions", &rand_actions);
    m.impl("rand_perms", &rand_perms);
    m.impl("spin_reset", &spin_reset);
    m.impl("spin_observation", &spin_observation);
    m.impl("spin_materialize", &spin_materialize);
    m.impl("rand_couplings", &rand_couplings);
    m.impl("spin_reset_dense", &spin_reset_dense);
    m.impl("spin_step_dense", &spin_step_dense);
    m.impl("spin_step", &spin_step);
    m.impl("mcpg_metro_rounds", &mcpg_metro_rounds);
    m.impl("mcpg_metro_stop", &mcpg_metro_stop);
    m.impl("mcpg_local_search", &mcpg_local_search);
    m.impl("mcpg_local_search_levels", &mcpg_local_search_levels);
    m.impl("mcpg_pick_best", &mcpg_pick_best);
    m.impl("mcpg_merge_best", &mcpg_merge_best);
    m.impl("mcpg_value_bit_sums", &mcpg_value_bit_sums);
    m.impl("mcpg_pack_chains", &mcpg_pack_chains);
    m.impl("mcpg_unpack_chains", &mcpg_unpack_chains);
    m.impl("qubo_local_search_value", &qubo_local_search_value);
    m.impl("qubo_sparse_local_search_value", &qubo_sparse_local_search_value);
    m.impl("tsp_tour_length", &tsp_tour_length);
    m.impl("tsp_swap_delta_all", &tsp_swap_delta_all);
    m.impl("tsp_apply_swap", &tsp_apply_swap);
    m.impl("tsp_2opt_delta", &tsp_2opt_delta);
    m.impl("tsp_2opt_best", &tsp_2opt_best);
    m.impl("isco_maxcut_step", &isco_maxcut_step);
    m.impl("isco_tsp_step", &isco_tsp_step);
}
