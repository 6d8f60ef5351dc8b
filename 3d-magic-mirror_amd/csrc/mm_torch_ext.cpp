// mm_torch_ext.cpp -- optional host-side fast path of the autograd API (DiffRender.render / recon_data / render_recon).
//
// The C ABI of libmm_render.so stays the boundary; this file is PLUMBING above it, compiled with the host compiler only (no device
// code, no HIP headers): one C++ call per autograd node allocates the outputs with ATen, fills the descriptor and enqueues the
// library's launches on the stream it is given -- what 3d-magic-mirror_amd/diff_render.py otherwise does with ~40 Python / ctypes
// statements per node (0.2-0.4 ms of host time per step against 0.12 ms of GPU time at B=48, 128x128).  Function addresses of the
// library come from the ctypes handle, so there is no link-time coupling and no second copy of the library.  If this module is not
// built, diff_render.py uses its Python path: same calls, same results.
#include <torch/extension.h>
#include <torch/csrc/autograd/custom_function.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // device guard + current stream of a ROCm build of torch (host headers only)
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "../../include/mm_render.h"

namespace {

// ---- DEFERRED FUSION (MMRenderDesc.fused_totals, ABI 6) ---------------------------------------------------------------------------------------
// The un-modified trainer calls `pred, att = render(...)` and later `recon_data(pred, gt)` (trainer.py:276,441).  When that `pred` is the untouched
// image of a render of this process, the loss VALUE is formed by mm_recon_data_forward as always, but its BACKWARD is routed into the render node:
// the render node hands out, next to the image, a one-element `token`; recon_data consumes (token, pred.detach(), gt) instead of (pred, gt); its
// backward returns dL/dloss as the token's gradient and launches nothing; the render node's backward, seeing a token gradient, runs
// mm_render_backward with fused_gt / fused_totals: no dL/drgba tensor, no recon_data backward launch, no read of it by the pixel pass -- and the
// bits mm_recon_data_backward + mm_render_backward would have produced (csrc/mm_pixel_bwd.hip: kDeferred).  Gradients that reach the image from
// its OTHER consumers arrive as grad_rgba as before and are added.
// What a render leaves for a later recon_data to find (keyed by the image's storage address), and what recon_data leaves for the render's backward:
struct Mailbox {
    const void* rgba_ptr = nullptr;
    const void* node = nullptr;             // the render's autograd node (identity of `pred`'s producer)
    uint32_t version = 0;                   // the image's version counter when render returned it
    int64_t B = 0, H = 0, W = 0;
    bool has_token = false;                 // the node declared a fifth output (the token's slot: output_nr 4).  The token TENSOR is not kept here: it would
                                            // close a reference cycle node -> context -> mailbox -> token -> grad_fn -> node that nothing ever frees;
                                            // recon_data makes a fresh tensor and hangs it on the node's output 4 (deferred_token below)
    bool claimed = false;                   // a recon_data has taken this render (a second one runs un-deferred)
    at::Tensor gt, recon_ws;                // recon_data's dense target and its workspace (the totals live in it)
    const float* totals = nullptr;
    double image_weight = 0.0;
};
std::mutex g_mail_mutex;
std::unordered_map<const void*, std::weak_ptr<Mailbox>> g_mail;      // image storage address -> its render's mailbox (weak: dies with the node)

// a 0-byte CPU tensor whose storage context owns a shared_ptr<Mailbox>: the form in which an AutogradContext can hold it (saved_data takes IValues)
void mailbox_deleter(void* ctx) { delete static_cast<std::shared_ptr<Mailbox>*>(ctx); }
at::Tensor mailbox_holder(const std::shared_ptr<Mailbox>& mb) {
    auto* ctx = new std::shared_ptr<Mailbox>(mb);
    c10::DataPtr dp(nullptr, ctx, &mailbox_deleter, c10::Device(c10::kCPU));
    c10::Storage st(c10::Storage::use_byte_size_t(), 0, std::move(dp), nullptr, false);
    return at::empty({0}, at::TensorOptions().dtype(at::kByte)).set_(st, 0, {0}, {1});
}
std::shared_ptr<Mailbox> mailbox_of(const at::Tensor& holder) {
    if (!holder.defined()) return nullptr;
    const c10::DataPtr& dp = holder.storage().data_ptr();
    if (dp.get_deleter() != &mailbox_deleter || !dp.get_context()) return nullptr;
    return *static_cast<std::shared_ptr<Mailbox>*>(dp.get_context());
}

typedef int (*render_fwd_t)(const MMRenderDesc*, void*);
typedef int (*render_bwd_t)(const MMRenderDesc*, const MMRenderGrads*, void*);
typedef int (*recon_t)(const MMReconDesc*, void*);
typedef size_t (*recon_ws_t)(const MMReconDesc*);

const float* fptr(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* mptr(at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
const float* optp(const c10::optional<at::Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr<float>() : nullptr; }

at::Tensor dense_f32(const at::Tensor& t, const c10::Device& dev, const char* what) {
    TORCH_CHECK(t.is_cuda(), "the MI355X render path needs tensors in device memory (got a ", t.device(), " tensor for ", what, "); there is no CPU fallback");
    if (t.scalar_type() == at::kFloat && t.device() == dev && t.is_contiguous()) return t.detach();
    return t.detach().to(dev, at::kFloat).contiguous();
}

void check(int rc, const char* what) { TORCH_CHECK(rc == MM_OK, what, " failed with status ", rc); }

// The stream a node's work is enqueued on is torch's CURRENT stream of the tensors' device at the time of the call, forward and backward
// alike (the autograd engine re-establishes the forward's stream around a node's backward; a caller driving autograd.grad under another
// stream context gets that one): never a raw handle remembered from the forward, which may be stale by then.  The guard makes the tensors'
// device current for the allocations and the launches.
typedef c10::hip::HIPGuardMasqueradingAsCUDA DeviceGuard;
int64_t current_stream(const c10::Device& dev) { return (int64_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream(); }

MMRenderDesc proto_desc(const std::string& proto) {
    TORCH_CHECK(proto.size() == sizeof(MMRenderDesc), "descriptor prototype of ", proto.size(), " bytes, expected ", sizeof(MMRenderDesc));
    MMRenderDesc d;
    std::memcpy(&d, proto.data(), sizeof d);
    return d;
}

// rgba (B,H,W,4), face_normals (B,F,3), imnormal (B,H,W,3 or empty), face_idx (B,H,W) int32, loss (scalar or undefined) + the dense inputs
std::vector<at::Tensor> render_forward(int64_t f_fwd, int64_t f_loss, const std::string& proto, at::Tensor vertices, at::Tensor textures,
                                       at::Tensor lights, c10::optional<at::Tensor> bg, at::Tensor azimuths, at::Tensor elevations,
                                       at::Tensor distances, at::Tensor biases, c10::optional<at::Tensor> gt, bool want_imnormal, double image_weight,
                                       at::Tensor ws, int64_t stream) {
    MMRenderDesc d = proto_desc(proto);
    const c10::Device dev = azimuths.device();
    vertices = dense_f32(vertices, dev, "vertices"); textures = dense_f32(textures, dev, "textures"); lights = dense_f32(lights, dev, "lights");
    azimuths = dense_f32(azimuths, dev, "azimuths").reshape({-1}); elevations = dense_f32(elevations, dev, "elevations").reshape({-1});
    distances = dense_f32(distances, dev, "distances").reshape({-1}); biases = dense_f32(biases, dev, "biases");
    at::Tensor bgt, gtt;
    if (bg.has_value() && bg->defined()) bgt = dense_f32(*bg, dev, "bg");
    const int64_t B = azimuths.size(0), H = d.H, W = d.W;
    TORCH_CHECK(B == d.B, "batch size ", B, " does not match the descriptor (", d.B, ")");
    TORCH_CHECK(vertices.dim() == 3 && vertices.size(0) == B && vertices.size(1) == d.V && vertices.size(2) == 3, "vertices must be (B,", d.V, ",3), got ", vertices.sizes());
    TORCH_CHECK(textures.dim() == 4 && textures.size(0) == B && textures.size(1) == 3 && textures.size(2) == d.Ht && textures.size(3) == d.Wt,
                "textures must be (B,3,Ht,Wt), got ", textures.sizes());
    TORCH_CHECK(lights.dim() == 2 && lights.size(0) == B && lights.size(1) == 9 && biases.dim() == 2 && biases.size(0) == B && biases.size(1) == 2 &&
                elevations.size(0) == B && distances.size(0) == B, "lights (B,9), biases (B,2), elevations/distances (B) expected");
    if (d.no_mask) TORCH_CHECK(bgt.defined() && bgt.dim() == 4 && bgt.size(0) == B && bgt.size(1) == 3 && bgt.size(2) == H && bgt.size(3) == W,
                               "bg must be (B,3,", H, ",", W, ")");
    auto opts = vertices.options();
    at::Tensor rgba = at::empty({B, H, W, 4}, opts), fn = at::empty({B, (int64_t)d.F, 3}, opts);
    at::Tensor face_idx = at::empty({B, H, W}, opts.dtype(at::kInt));
    at::Tensor imn = want_imnormal ? at::empty({B, H, W, 3}, opts) : at::empty({0}, opts);
    at::Tensor loss;
    d.vertices = fptr(vertices); d.textures = fptr(textures); d.lights = fptr(lights); d.bg = d.no_mask ? fptr(bgt) : nullptr;
    d.azimuths = fptr(azimuths); d.elevations = fptr(elevations); d.distances = fptr(distances); d.biases = fptr(biases);
    d.rgba = mptr(rgba); d.face_idx = face_idx.data_ptr<int32_t>(); d.face_normals = mptr(fn); d.imnormal = want_imnormal ? mptr(imn) : nullptr;
    d.workspace = ws.data_ptr(); d.workspace_bytes = (size_t)ws.numel();
    if (gt.has_value() && gt->defined()) {
        gtt = dense_f32(*gt, dev, "gt_data");
        TORCH_CHECK(gtt.dim() == 4 && gtt.size(0) == B && gtt.size(1) == 4 && gtt.size(2) == H && gtt.size(3) == W, "gt_data must be (B,4,", H, ",", W, "), got ", gtt.sizes());
        loss = at::empty({}, opts);
        d.fused_gt = fptr(gtt); d.fused_image_weight = (float)image_weight; d.fused_loss = mptr(loss);
    }
    check(((render_fwd_t)f_fwd)(&d, (void*)stream), "mm_render_forward");
    if (gtt.defined()) check(((render_fwd_t)f_loss)(&d, (void*)stream), "mm_render_fused_loss");
    return {rgba, fn, imn, face_idx, loss, vertices, textures, lights, bgt, azimuths, elevations, distances, biases, gtt};
}

// gradients of vertices, textures, lights, bg (undefined unless no_mask), azimuths, elevations, distances, biases
std::vector<at::Tensor> render_backward(int64_t f_bwd, const std::string& proto, at::Tensor vertices, at::Tensor textures, at::Tensor lights,
                                        c10::optional<at::Tensor> bg, at::Tensor azimuths, at::Tensor elevations, at::Tensor distances,
                                        at::Tensor biases, at::Tensor face_idx, at::Tensor fn, c10::optional<at::Tensor> gt,
                                        c10::optional<at::Tensor> rgba_fwd, c10::optional<at::Tensor> g_rgba, c10::optional<at::Tensor> g_fn,
                                        c10::optional<at::Tensor> g_loss, double image_weight, at::Tensor ws, int64_t stream, int64_t deferred_totals) {
    MMRenderDesc d = proto_desc(proto);
    const int64_t B = azimuths.size(0), H = d.H, W = d.W;
    const bool fused = gt.has_value() && gt->defined();
    const bool deferred = fused && deferred_totals != 0;         // deferred fusion: gt / totals come from the recon_data that consumed this render's image
    at::Tensor grgba, gfn, gloss;
    d.vertices = fptr(vertices); d.textures = fptr(textures); d.lights = fptr(lights); d.bg = d.no_mask ? optp(bg) : nullptr;
    d.azimuths = fptr(azimuths); d.elevations = fptr(elevations); d.distances = fptr(distances); d.biases = fptr(biases);
    d.face_idx = face_idx.data_ptr<int32_t>(); d.face_normals = mptr(fn); d.imnormal = nullptr;
    d.workspace = ws.data_ptr(); d.workspace_bytes = (size_t)ws.numel();
    if (g_fn.has_value() && g_fn->defined()) gfn = g_fn->to(at::kFloat).contiguous();
    if (fused) {
        // An undefined gradient of the loss output means the loss took no part in what is being differentiated (materialize_grads is off):
        // its gradient is ZERO, never one -- e.g. reg.backward() through attributes['face_normals'] only.
        gloss = (g_loss.has_value() && g_loss->defined()) ? g_loss->to(at::kFloat).reshape({}).contiguous()
                                                          : at::zeros({}, vertices.options().dtype(at::kFloat));
        d.fused_gt = optp(gt); d.fused_image_weight = (float)image_weight; d.fused_grad_loss = fptr(gloss);
        d.rgba = nullptr;                                    // the backward re-forms the prediction per pixel: the image is not read back (nor saved)
        if (deferred) {
            d.fused_totals = reinterpret_cast<const float*>(deferred_totals); d.fused_contour = 0.f;
            if (g_rgba.has_value() && g_rgba->defined()) grgba = g_rgba->to(at::kFloat).contiguous();   // the image's other consumers: added by the kernel
        }
    } else {
        grgba = (g_rgba.has_value() && g_rgba->defined()) ? g_rgba->to(at::kFloat).contiguous() : at::zeros({B, H, W, 4}, vertices.options());
        d.rgba = nullptr;                                    // (not read by the backward)
    }
    at::Tensor gv = at::empty_like(vertices), gt_ = at::empty_like(textures), gl = at::empty_like(lights), gbg;
    if (d.no_mask) gbg = at::empty_like(*bg);
    at::Tensor ga = at::empty_like(azimuths), ge = at::empty_like(elevations), gd = at::empty_like(distances), gb = at::empty_like(biases);
    MMRenderGrads g;
    g.grad_rgba = (fused && !deferred) ? nullptr : fptr(grgba); g.grad_face_normals = fptr(gfn); g.grad_vertices = mptr(gv); g.grad_textures = mptr(gt_);
    g.grad_lights = mptr(gl); g.grad_bg = mptr(gbg); g.grad_azimuths = mptr(ga); g.grad_elevations = mptr(ge); g.grad_distances = mptr(gd);
    g.grad_biases = mptr(gb);
    check(((render_bwd_t)f_bwd)(&d, &g, (void*)stream), "mm_render_backward");
    return {gv, gt_, gl, gbg, ga, ge, gd, gb};
}

// loss, dense prediction, dense target, workspace
std::vector<at::Tensor> recon_forward(int64_t f_ws, int64_t f_fwd, at::Tensor pred, at::Tensor gt, double image_weight, double contour, int64_t stream) {
    TORCH_CHECK(pred.is_cuda() && gt.is_cuda(), "the MI355X render path needs tensors in device memory; there is no CPU fallback");
    const c10::Device dev = pred.device();
    pred = pred.detach().to(at::kFloat);
    if (!pred.is_non_overlapping_and_dense()) pred = pred.contiguous();
    gt = gt.detach().to(dev, at::kFloat).contiguous();
    TORCH_CHECK(pred.dim() == 4 && pred.size(1) == 4 && gt.sizes() == pred.sizes(), "recon_data expects (B,4,H,W) prediction and target, got ", pred.sizes(), " / ", gt.sizes());
    at::Tensor loss = at::empty({}, pred.options());
    MMReconDesc d;
    std::memset(&d, 0, sizeof d);
    d.B = (int32_t)pred.size(0); d.H = (int32_t)pred.size(2); d.W = (int32_t)pred.size(3);
    d.pred = fptr(pred); d.gt = fptr(gt);
    for (int i = 0; i < 4; ++i) d.pred_strides[i] = pred.stride(i);
    d.image_weight = (float)image_weight; d.contour = (float)contour; d.loss = mptr(loss);
    at::Tensor ws = at::empty({(int64_t)((recon_ws_t)f_ws)(&d)}, pred.options().dtype(at::kByte));
    d.workspace = ws.data_ptr(); d.workspace_bytes = (size_t)ws.numel();
    check(((recon_t)f_fwd)(&d, (void*)stream), "mm_recon_data_forward");
    return {loss, pred, gt, ws};
}

at::Tensor recon_backward(int64_t f_bwd, at::Tensor pred, at::Tensor gt, at::Tensor ws, at::Tensor g_loss, double image_weight, double contour, int64_t stream) {
    g_loss = g_loss.to(pred.device(), at::kFloat).contiguous();
    at::Tensor grad = at::empty_strided(pred.sizes(), pred.strides(), pred.options());
    MMReconDesc d;
    std::memset(&d, 0, sizeof d);
    d.B = (int32_t)pred.size(0); d.H = (int32_t)pred.size(2); d.W = (int32_t)pred.size(3);
    d.pred = fptr(pred); d.gt = fptr(gt);
    for (int i = 0; i < 4; ++i) d.pred_strides[i] = pred.stride(i);
    d.image_weight = (float)image_weight; d.contour = (float)contour;
    d.grad_loss = fptr(g_loss); d.grad_pred = mptr(grad);
    d.workspace = ws.data_ptr(); d.workspace_bytes = (size_t)ws.numel();
    check(((recon_t)f_bwd)(&d, (void*)stream), "mm_recon_data_backward");
    return grad;
}

// ---- the autograd nodes themselves, in C++: no Python (and no GIL hand-over to the autograd thread) in the backward -----------------------
// The stream is captured at forward time: the autograd engine runs a node's backward under the stream its forward ran on.
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

class RenderNode : public torch::autograd::Function<RenderNode> {
 public:
    static tensor_list forward(AutogradContext* ctx, int64_t f_fwd, int64_t f_loss, int64_t f_bwd, std::string proto, int64_t ws_bytes,
                               at::Tensor vertices, at::Tensor textures, at::Tensor lights, c10::optional<at::Tensor> bg, at::Tensor azimuths,
                               at::Tensor elevations, at::Tensor distances, at::Tensor biases, c10::optional<at::Tensor> gt, bool want_imnormal,
                               double image_weight, int64_t stream, bool defer) {
        TORCH_CHECK(azimuths.is_cuda(), "the MI355X render path needs tensors in device memory; there is no CPU fallback");
        const DeviceGuard guard(azimuths.device());
        stream = current_stream(azimuths.device());              // (the argument is kept for the binding's signature only)
        // the camera scalars may arrive as (B), (B,1), ...: the kernels see (B), the gradients go back in the caller's shapes (advisor r05)
        ctx->saved_data["shape_a"] = azimuths.sizes().vec(); ctx->saved_data["shape_e"] = elevations.sizes().vec(); ctx->saved_data["shape_d"] = distances.sizes().vec();
        at::Tensor ws = at::empty({ws_bytes}, azimuths.options().dtype(at::kByte));   // (the caching allocator is the workspace pool)
        auto out = render_forward(f_fwd, f_loss, proto, vertices, textures, lights, bg, azimuths, elevations, distances, biases, gt, want_imnormal,
                                  image_weight, ws, stream);
        const bool fused = out[13].defined();
        ctx->saved_data["f_bwd"] = f_bwd; ctx->saved_data["proto"] = proto;
        ctx->saved_data["image_weight"] = image_weight; ctx->saved_data["fused"] = fused;
        // dense inputs, forward products the backward re-reads, and the workspace (alive until this node dies)
        ctx->save_for_backward({out[5], out[6], out[7], out[8], out[9], out[10], out[11], out[12], out[3], out[1], out[13],
                                at::Tensor(), ws});          // (the image is not saved: the caller may overwrite it)
        ctx->mark_non_differentiable({out[3], out[2]});
        ctx->set_materialize_grads(false);                       // an output nobody used arrives as an undefined gradient, not as a zero-filled tensor (three
                                                                 // allocations + fill launches per step when only the image feeds the loss)
        if (fused) ctx->mark_non_differentiable({out[0]});
        tensor_list ret = {out[0], out[1], out[2], out[3]};
        if (fused) ret.push_back(out[4]);
        else if (defer) {
            // deferred fusion: a fifth output whose only job is to carry dL/dloss of a later recon_data(image, gt) back into THIS node (never read, never
            // written: no launch), and the mailbox that recon_data finds through the image's address
            auto mb = std::make_shared<Mailbox>();
            mb->rgba_ptr = out[0].data_ptr(); mb->B = out[0].size(0); mb->H = out[0].size(1); mb->W = out[0].size(2);
            mb->has_token = true;
            ctx->saved_data["mb"] = mailbox_holder(mb);
            { std::lock_guard<std::mutex> lock(g_mail_mutex);
              if (g_mail.size() > 256) for (auto it = g_mail.begin(); it != g_mail.end();) it = it->second.expired() ? g_mail.erase(it) : std::next(it);
              g_mail[mb->rgba_ptr] = mb; }
            ret.push_back(at::empty({1}, out[0].options()));    // (declares output 4 and its metadata; dropped by render_node)
        }
        return ret;
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list g) {
        const auto sv = ctx->get_saved_variables();
        const bool fused = ctx->saved_data["fused"].toBool();
        const DeviceGuard guard(sv[4].device());
        auto opt = [](const at::Tensor& t) { return t.defined() ? c10::optional<at::Tensor>(t) : c10::nullopt; };
        // deferred fusion: a recon_data consumed this render's image and its loss takes part in what is differentiated (the token has a gradient)
        std::shared_ptr<Mailbox> mb;
        if (!fused && g.size() > 4 && g[4].defined() && ctx->saved_data.count("mb")) mb = mailbox_of(ctx->saved_data["mb"].toTensor());
        const bool deferred = mb && mb->gt.defined() && mb->totals != nullptr;
        TORCH_CHECK(deferred || fused || g.size() <= 4 || !g[4].defined(), "a recon_data token carries a gradient but its render has no recon_data on record");
        auto gr = deferred
            ? render_backward(ctx->saved_data["f_bwd"].toInt(), ctx->saved_data["proto"].toStringRef(), sv[0], sv[1], sv[2], opt(sv[3]), sv[4], sv[5],
                              sv[6], sv[7], sv[8], sv[9], opt(mb->gt), c10::nullopt, opt(g[0]), opt(g[1]), opt(g[4]), mb->image_weight, sv[12],
                              current_stream(sv[4].device()), reinterpret_cast<int64_t>(mb->totals))
            : render_backward(ctx->saved_data["f_bwd"].toInt(), ctx->saved_data["proto"].toStringRef(), sv[0], sv[1], sv[2], opt(sv[3]), sv[4], sv[5],
                              sv[6], sv[7], sv[8], sv[9], opt(sv[10]), opt(sv[11]), opt(g[0]), opt(g[1]),
                              (fused && g.size() > 4) ? opt(g[4]) : c10::nullopt, ctx->saved_data["image_weight"].toDouble(), sv[12],
                              current_stream(sv[4].device()), 0);
        // one entry per forward argument: five non-tensors, then vertices, textures, lights, bg, azimuths, elevations, distances, biases, ...
        return {at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), gr[0], gr[1], gr[2], gr[3],
                gr[4].reshape(ctx->saved_data["shape_a"].toIntVector()), gr[5].reshape(ctx->saved_data["shape_e"].toIntVector()),
                gr[6].reshape(ctx->saved_data["shape_d"].toIntVector()), gr[7],
                at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// recon_data on the untouched image of a render that handed out a token (deferred fusion, top of this file): the forward is mm_recon_data_forward on
// the image as always (same launches, same value); the backward launches NOTHING -- dL/dloss travels to the render node as the token's gradient.
class ReconDeferredNode : public torch::autograd::Function<ReconDeferredNode> {
 public:
    static at::Tensor forward(AutogradContext* ctx, at::Tensor token, at::Tensor holder, int64_t f_ws, int64_t f_fwd, int64_t f_tot, at::Tensor pred,
                              at::Tensor gt, double image_weight) {
        const DeviceGuard guard(pred.device());
        (void)token;
        auto out = recon_forward(f_ws, f_fwd, pred, gt, image_weight, 0.0, current_stream(pred.device()));
        auto mb = mailbox_of(holder);
        TORCH_CHECK(mb, "deferred recon_data without its render's mailbox");
        MMReconDesc d;
        std::memset(&d, 0, sizeof d);
        d.B = (int32_t)out[1].size(0); d.H = (int32_t)out[1].size(2); d.W = (int32_t)out[1].size(3);
        d.workspace = out[3].data_ptr(); d.workspace_bytes = (size_t)out[3].numel();
        typedef const float* (*totals_t)(const MMReconDesc*);
        mb->totals = ((totals_t)f_tot)(&d);
        TORCH_CHECK(mb->totals != nullptr, "mm_recon_data_totals failed");
        mb->gt = out[2]; mb->recon_ws = out[3]; mb->image_weight = image_weight;     // alive as long as the render node is
        ctx->set_materialize_grads(false);
        return out[0];
    }
    static tensor_list backward(AutogradContext* ctx, tensor_list g) {
        (void)ctx;
        at::Tensor gt;                                           // the token's gradient = dL/dloss (one float on the device; a view where the caller's is one)
        if (g[0].defined()) gt = g[0].to(at::kFloat).reshape({1});
        return {gt, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

class ReconNode : public torch::autograd::Function<ReconNode> {
 public:
    static at::Tensor forward(AutogradContext* ctx, int64_t f_ws, int64_t f_fwd, int64_t f_bwd, at::Tensor pred, at::Tensor gt, double image_weight,
                              double contour, int64_t stream) {
        TORCH_CHECK(pred.is_cuda(), "the MI355X render path needs tensors in device memory; there is no CPU fallback");
        const DeviceGuard guard(pred.device());
        auto out = recon_forward(f_ws, f_fwd, pred, gt, image_weight, contour, current_stream(pred.device()));
        (void)stream;
        ctx->saved_data["f_bwd"] = f_bwd; ctx->saved_data["image_weight"] = image_weight; ctx->saved_data["contour"] = contour;
        ctx->save_for_backward({out[1], out[2], out[3]});
        return out[0];
    }
    static tensor_list backward(AutogradContext* ctx, tensor_list g) {
        const auto sv = ctx->get_saved_variables();
        const DeviceGuard guard(sv[0].device());
        at::Tensor grad = recon_backward(ctx->saved_data["f_bwd"].toInt(), sv[0], sv[1], sv[2], g[0], ctx->saved_data["image_weight"].toDouble(),
                                         ctx->saved_data["contour"].toDouble(), current_stream(sv[0].device()));
        return {at::Tensor(), at::Tensor(), at::Tensor(), grad, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// DiffRender.render_geometry (MMRenderDesc.geometry_only: the vertex stage alone, for the render whose image trainer.py:367 discards):
// face_normals (B,F,3) with a backward to vertices and the four camera inputs.  The prototype carries geometry_only = 1.
class GeometryNode : public torch::autograd::Function<GeometryNode> {
 public:
    static at::Tensor forward(AutogradContext* ctx, int64_t f_fwd, int64_t f_bwd, std::string proto, int64_t ws_bytes, at::Tensor vertices,
                              at::Tensor azimuths, at::Tensor elevations, at::Tensor distances, at::Tensor biases) {
        TORCH_CHECK(azimuths.is_cuda(), "the MI355X render path needs tensors in device memory; there is no CPU fallback");
        const c10::Device dev = azimuths.device();
        const DeviceGuard guard(dev);
        MMRenderDesc d = proto_desc(proto);
        TORCH_CHECK(d.geometry_only == 1, "GeometryNode needs a geometry-only descriptor prototype");
        vertices = dense_f32(vertices, dev, "vertices");
        ctx->saved_data["shape_a"] = azimuths.sizes().vec(); ctx->saved_data["shape_e"] = elevations.sizes().vec(); ctx->saved_data["shape_d"] = distances.sizes().vec();
        azimuths = dense_f32(azimuths, dev, "azimuths").reshape({-1}); elevations = dense_f32(elevations, dev, "elevations").reshape({-1});
        distances = dense_f32(distances, dev, "distances").reshape({-1}); biases = dense_f32(biases, dev, "biases");
        const int64_t B = azimuths.size(0);
        TORCH_CHECK(B == d.B, "batch size ", B, " does not match the descriptor (", d.B, ")");
        TORCH_CHECK(vertices.dim() == 3 && vertices.size(0) == B && vertices.size(1) == d.V && vertices.size(2) == 3, "vertices must be (B,", d.V, ",3), got ", vertices.sizes());
        TORCH_CHECK(biases.dim() == 2 && biases.size(0) == B && biases.size(1) == 2 && elevations.size(0) == B && distances.size(0) == B,
                    "biases (B,2), elevations/distances (B) expected");
        at::Tensor ws = at::empty({ws_bytes}, azimuths.options().dtype(at::kByte));
        at::Tensor fn = at::empty({B, (int64_t)d.F, 3}, vertices.options());
        d.vertices = fptr(vertices); d.azimuths = fptr(azimuths); d.elevations = fptr(elevations); d.distances = fptr(distances); d.biases = fptr(biases);
        d.face_normals = mptr(fn); d.workspace = ws.data_ptr(); d.workspace_bytes = (size_t)ws.numel();
        check(((render_fwd_t)f_fwd)(&d, (void*)current_stream(dev)), "mm_render_forward (geometry only)");
        ctx->saved_data["f_bwd"] = f_bwd; ctx->saved_data["proto"] = proto;
        ctx->save_for_backward({vertices, azimuths, elevations, distances, biases, ws});
        ctx->set_materialize_grads(false);
        return fn;
    }
    static tensor_list backward(AutogradContext* ctx, tensor_list g) {
        const auto sv = ctx->get_saved_variables();
        tensor_list none(9);
        if (!g[0].defined()) return none;                        // face_normals took no part in what is differentiated
        const DeviceGuard guard(sv[1].device());
        MMRenderDesc d = proto_desc(ctx->saved_data["proto"].toStringRef());
        at::Tensor gfn = g[0].to(at::kFloat).contiguous();
        at::Tensor ws = sv[5];
        d.vertices = fptr(sv[0]); d.azimuths = fptr(sv[1]); d.elevations = fptr(sv[2]); d.distances = fptr(sv[3]); d.biases = fptr(sv[4]);
        d.face_normals = mptr(gfn);                              // (a valid pointer for the argument check; the backward does not read the normals)
        d.workspace = ws.data_ptr(); d.workspace_bytes = (size_t)ws.numel();
        at::Tensor gv = at::empty_like(sv[0]), ga = at::empty_like(sv[1]), ge = at::empty_like(sv[2]), gd = at::empty_like(sv[3]), gb = at::empty_like(sv[4]);
        MMRenderGrads gr;
        std::memset(&gr, 0, sizeof gr);
        gr.grad_face_normals = fptr(gfn); gr.grad_vertices = mptr(gv); gr.grad_azimuths = mptr(ga); gr.grad_elevations = mptr(ge);
        gr.grad_distances = mptr(gd); gr.grad_biases = mptr(gb);
        check(((render_bwd_t)ctx->saved_data["f_bwd"].toInt())(&d, &gr, (void*)current_stream(sv[1].device())), "mm_render_backward (geometry only)");
        none[4] = gv; none[5] = ga.reshape(ctx->saved_data["shape_a"].toIntVector()); none[6] = ge.reshape(ctx->saved_data["shape_e"].toIntVector());
        none[7] = gd.reshape(ctx->saved_data["shape_d"].toIntVector()); none[8] = gb;
        return none;
    }
};

at::Tensor geometry_node(int64_t f_fwd, int64_t f_bwd, std::string proto, int64_t ws_bytes, at::Tensor vertices, at::Tensor azimuths, at::Tensor elevations,
                         at::Tensor distances, at::Tensor biases) {
    return GeometryNode::apply(f_fwd, f_bwd, proto, ws_bytes, vertices, azimuths, elevations, distances, biases);
}

// defer: hand out the token a later recon_data(image, gt) can route its backward through (deferred fusion; the returned list is still
// {rgba, face_normals, imnormal, face_idx}: the token lives in the mailbox)
tensor_list render_node(int64_t f_fwd, int64_t f_loss, int64_t f_bwd, std::string proto, int64_t ws_bytes, at::Tensor vertices, at::Tensor textures,
                        at::Tensor lights, c10::optional<at::Tensor> bg, at::Tensor azimuths, at::Tensor elevations, at::Tensor distances, at::Tensor biases,
                        c10::optional<at::Tensor> gt, bool want_imnormal, double image_weight, int64_t stream, bool defer) {
    const bool fused = gt.has_value() && gt->defined();
    defer = defer && !fused && at::GradMode::is_enabled();
    tensor_list out = RenderNode::apply(f_fwd, f_loss, f_bwd, proto, ws_bytes, vertices, textures, lights, bg, azimuths, elevations, distances, biases, gt,
                                        want_imnormal, image_weight, stream, defer);
    if (defer && out.size() > 4) {
        std::shared_ptr<Mailbox> mb;
        { std::lock_guard<std::mutex> lock(g_mail_mutex);
          auto it = g_mail.find(out[0].data_ptr());
          if (it != g_mail.end()) mb = it->second.lock(); }
        if (mb && out[0].grad_fn() && out[4].grad_fn().get() == out[0].grad_fn().get() && out[4].output_nr() == 4) {
            mb->node = out[0].grad_fn().get(); mb->version = out[0]._version();
        } else if (mb) mb->claimed = true;                       // nothing requires grad: no backward will ever run, nothing to defer
        out.pop_back();
    }
    return out;
}

// The render whose image `pred` is, if recon_data may route its backward through it: pred is the (B,4,H,W) permute view of (or the NHWC image itself,
// seen as (B,4,H,W)) output 0 of a render node that handed out a token, float32, never modified in place since, not yet taken by another recon_data.
std::shared_ptr<Mailbox> deferrable_render(const at::Tensor& pred) {
    if (!pred.defined() || !pred.is_cuda() || pred.scalar_type() != at::kFloat || pred.dim() != 4 || !pred.requires_grad() || !at::GradMode::is_enabled()) return nullptr;
    std::shared_ptr<Mailbox> mb;
    { std::lock_guard<std::mutex> lock(g_mail_mutex);
      auto it = g_mail.find(pred.data_ptr());
      if (it != g_mail.end()) mb = it->second.lock(); }
    if (!mb || mb->claimed || !mb->node || !mb->has_token) return nullptr;
    if (pred.size(0) != mb->B || pred.size(1) != 4 || pred.size(2) != mb->H || pred.size(3) != mb->W) return nullptr;
    if (pred.stride(0) != 4 * mb->H * mb->W || pred.stride(1) != 1 || pred.stride(2) != 4 * mb->W || pred.stride(3) != 4) return nullptr;
    if (pred._version() != mb->version) return nullptr;         // written in place since the render returned it
    const auto fn = pred.grad_fn();
    if (!fn) return nullptr;
    // `pred` must BE the render's image: one permute away from output 0 of the node (what DiffRender.render returns)
    if (fn->num_inputs() < 1 || fn->next_edges().size() != 1) return nullptr;
    const auto& e = fn->next_edge(0);
    if (e.function.get() != mb->node || e.input_nr != 0 || fn->name().find("Permute") == std::string::npos) return nullptr;
    return mb;
}

// a one-element tensor that IS output 4 of `pred`'s render node as far as autograd is concerned: whatever gradient reaches it arrives in that node's
// backward as g[4].  (`pred` has passed deferrable_render: its grad_fn is the permute whose only input edge is the node's output 0.)
at::Tensor deferred_token(const at::Tensor& pred) {
    at::Tensor tok = at::empty({1}, pred.options());
    torch::autograd::impl::set_gradient_edge(tok, torch::autograd::Edge(pred.grad_fn()->next_edge(0).function, 4));
    return tok;
}

at::Tensor recon_node(int64_t f_ws, int64_t f_fwd, int64_t f_bwd, at::Tensor pred, at::Tensor gt, double image_weight, double contour, int64_t stream,
                      int64_t f_tot, bool allow_defer) {
    if (allow_defer && f_tot != 0 && !(contour > 0.0)) {
        if (auto mb = deferrable_render(pred)) {
            mb->claimed = true;
            // (find the holder again through the node's context is not possible from here: a second holder of the same mailbox travels as an argument)
            return ReconDeferredNode::apply(deferred_token(pred), mailbox_holder(mb), f_ws, f_fwd, f_tot, pred.detach(), gt, image_weight);
        }
    }
    return ReconNode::apply(f_ws, f_fwd, f_bwd, pred, gt, image_weight, contour, stream);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("render_forward", &render_forward);
    m.def("render_backward", &render_backward);
    m.def("recon_forward", &recon_forward);
    m.def("recon_backward", &recon_backward);
    m.def("render", &render_node);
    m.def("recon_data", &recon_node);
    m.def("render_geometry", &geometry_node);
    m.def("desc_bytes", []() { return (int64_t)sizeof(MMRenderDesc); });
    m.def("deferrable", [](at::Tensor pred) { return deferrable_render(pred) != nullptr; });   // tests / diagnostics: would recon_data(pred, .) defer?
}
