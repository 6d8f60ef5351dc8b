// mm_abi.hip -- extern "C" entry points of libmm_render.so (declared in include/mm_render.h): argument validation,
// workspace carving and launch sequencing.  No allocation, no host synchronisation, no global state.
#include <vector>
#include <cstdio>
#include <cstdlib>
#include "mm_device.h"

namespace mm {
int launch_vertex_fwd(const MMRenderDesc*, const Workspace&, hipStream_t);
int launch_vertex_bwd(const MMRenderDesc*, const MMRenderGrads*, const Workspace&, hipStream_t);
int launch_raster_fwd(const MMRenderDesc*, const Workspace&, hipStream_t);
int launch_raster_bwd(const MMRenderDesc*, const MMRenderGrads*, const Workspace&, hipStream_t);
int launch_fused_loss(const MMRenderDesc*, const Workspace&, hipStream_t);
size_t recon_workspace_bytes(const MMReconDesc*);
int launch_recon_fwd(const MMReconDesc*, hipStream_t);
int launch_recon_bwd(const MMReconDesc*, hipStream_t);
const float* recon_totals(const MMReconDesc*);
int launch_nn(int, int, int, const float*, const float*, float*, int32_t*, hipStream_t);
int launch_nn_both(int, int, int, const float*, const float*, float*, int32_t*, float*, int32_t*, hipStream_t);
size_t reg_workspace_bytes(const MMMeshRegDesc*);
int launch_reg_fwd(const MMMeshRegDesc*, hipStream_t);
int launch_reg_bwd(const MMMeshRegDesc*, const MMMeshRegGrads*, hipStream_t);
size_t att_workspace_bytes(const MMAttLossDesc*);
int launch_att_fwd(const MMAttLossDesc*, hipStream_t);
int launch_att_bwd(const MMAttLossDesc*, const MMAttLossGrads*, hipStream_t);
int launch_texflow_fwd(const MMTexFlowDesc*, hipStream_t);
int launch_texflow_bwd(const MMTexFlowDesc*, const MMTexFlowGrads*, hipStream_t);
}  // namespace mm

static int check_render(const MMRenderDesc* d, bool backward) {
    if (!d) return MM_ERR_NULL_POINTER;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->V <= 0 || d->F <= 0 || d->Ht <= 0 || d->Wt <= 0) return MM_ERR_BAD_SHAPE;
    if (d->knum <= 0) return MM_ERR_UNSUPPORTED;
    if (d->H > 65535 || d->W > 65535) return MM_ERR_UNSUPPORTED;                 // pixel boxes are packed in 16 + 16 bits
    if (d->geometry_only) {                                                       // vertex stage only: what it reads and writes
        if (!d->faces || !d->vertices || !d->azimuths || !d->elevations || !d->distances || !d->biases || !d->face_normals) return MM_ERR_NULL_POINTER;
    } else {
        if (!d->faces || !d->face_uvs || !d->vertices || !d->textures || !d->lights || !d->azimuths || !d->elevations ||
            !d->distances || !d->biases || !d->face_idx || !d->face_normals)
            return MM_ERR_NULL_POINTER;
        if (!backward && !d->rgba) return MM_ERR_NULL_POINTER;                    // (the backward never reads the image: rgba may be NULL there)
        if (d->no_mask && !d->bg) return MM_ERR_NULL_POINTER;
    }
    if (d->fused_gt && !d->geometry_only) {                                       // the contour term of the fused loss (include/mm_render.h)
        if (!(d->fused_contour >= 0.f)) return MM_ERR_BAD_SHAPE;
        if (d->fused_contour > 0.f && ((d->H & 3) || (d->W & 3))) return MM_ERR_BAD_SHAPE;
    }
    if (d->fused_totals) {                                                        // deferred fusion: the backward of a render whose recon_data ran on its own
        if (!backward || d->geometry_only) return MM_ERR_UNSUPPORTED;
        if (!d->fused_gt) return MM_ERR_NULL_POINTER;
        if (d->fused_contour != 0.f) return MM_ERR_UNSUPPORTED;
    }
    if (backward && !d->vc_table) return MM_ERR_NULL_POINTER;
    if (backward && d->vc_stride <= 0) return MM_ERR_BAD_SHAPE;
    if (!d->workspace || d->workspace_bytes < mm_query_workspace(d) || ((uintptr_t)d->workspace & 255)) return MM_ERR_WORKSPACE;
    return MM_OK;
}

extern "C" {

size_t mm_query_workspace(const MMRenderDesc* d) {
    if (!d || d->B <= 0 || d->V <= 0 || d->F <= 0 || d->H <= 0 || d->W <= 0 || d->Ht <= 0 || d->Wt <= 0) return 0;
    return mm::carve_workspace(nullptr, d->B, d->V, d->F, d->H, d->W, d->Ht, d->Wt, 0, d->geometry_only != 0).bytes;
}

int mm_render_forward(const MMRenderDesc* d, mm_stream_t stream) {
    int st = check_render(d, false);
    if (st != MM_OK) return st;
    const mm::Workspace w = mm::carve_workspace(d->workspace, d->B, d->V, d->F, d->H, d->W, d->Ht, d->Wt, d->workspace_bytes, d->geometry_only != 0);
    hipStream_t s = (hipStream_t)stream;
    mm::clear_stale_error();
    st = mm::launch_vertex_fwd(d, w, s);
    if (st != MM_OK || d->geometry_only) return st;
    return mm::launch_raster_fwd(d, w, s);      // tile order + raster
}

int mm_render_fused_loss(const MMRenderDesc* d, mm_stream_t stream) {
    int st = check_render(d, false);
    if (st != MM_OK) return st;
    if (!d->fused_gt || !d->fused_loss) return MM_ERR_NULL_POINTER;
    const mm::Workspace w = mm::carve_workspace(d->workspace, d->B, d->V, d->F, d->H, d->W, d->Ht, d->Wt, d->workspace_bytes, d->geometry_only != 0);
    mm::clear_stale_error();
    return mm::launch_fused_loss(d, w, (hipStream_t)stream);
}

int mm_debug_workspace_layout(const MMRenderDesc* d, size_t* out5) {
    if (!d || !out5) return MM_ERR_NULL_POINTER;
    const mm::Workspace w = mm::carve_workspace(nullptr, d->B, d->V, d->F, d->H, d->W, d->Ht, d->Wt);
    out5[0] = (size_t)((char*)w.chunkmap - (char*)nullptr); out5[1] = (size_t)((char*)w.items - (char*)nullptr);
    out5[2] = (size_t)((char*)w.nitems - (char*)nullptr); out5[3] = (size_t)((char*)w.part - (char*)nullptr); out5[4] = (size_t)w.item_cap;
    out5[5] = (size_t)((char*)w.gp - (char*)nullptr); out5[6] = (size_t)((char*)w.gp2 - (char*)nullptr); out5[7] = (size_t)((char*)w.soft - (char*)nullptr);
    out5[8] = (size_t)((char*)w.tcnt - (char*)nullptr); out5[9] = (size_t)w.ntiles; out5[10] = (size_t)w.trcap; out5[11] = (size_t)((char*)w.trcnt - (char*)nullptr);
    return MM_OK;
}

int mm_render_status(const MMRenderDesc* d, mm_stream_t stream, int32_t* dropped_host) {
    if (!d) return MM_ERR_NULL_POINTER;                            // (only the shape and the workspace are looked at)
    if (d->B <= 0 || d->V <= 0 || d->F <= 0 || d->H <= 0 || d->W <= 0 || d->Ht <= 0 || d->Wt <= 0) return MM_ERR_BAD_SHAPE;
    if (!d->workspace) return MM_ERR_NULL_POINTER;
    if (d->workspace_bytes < mm_query_workspace(d) || ((uintptr_t)d->workspace & 255)) return MM_ERR_BAD_SHAPE;   // (MM_ERR_WORKSPACE is this call's "records were dropped")
    const mm::Workspace w = mm::carve_workspace(d->workspace, d->B, d->V, d->F, d->H, d->W, d->Ht, d->Wt, d->workspace_bytes, d->geometry_only != 0);
    std::vector<int32_t> h((size_t)d->B);
    if (hipMemcpyAsync(h.data(), w.tstatus, (size_t)d->B * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return MM_ERR_LAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return MM_ERR_LAUNCH;
    bool any = false;
    for (int b = 0; b < d->B; ++b) { any = any || h[b] != 0; if (dropped_host) dropped_host[b] = h[b]; }
    return any ? MM_ERR_WORKSPACE : MM_OK;
}

int mm_render_backward(const MMRenderDesc* d, const MMRenderGrads* g, mm_stream_t stream) {
    int st = check_render(d, true);
    if (st != MM_OK) return st;
    if (!g || !g->grad_vertices || !g->grad_azimuths || !g->grad_elevations || !g->grad_distances || !g->grad_biases) return MM_ERR_NULL_POINTER;
    const mm::Workspace w = mm::carve_workspace(d->workspace, d->B, d->V, d->F, d->H, d->W, d->Ht, d->Wt, d->workspace_bytes, d->geometry_only != 0);
    hipStream_t s = (hipStream_t)stream;
    if (d->geometry_only) {                                       // nothing was rasterised: the gradient arrives through face_normals alone
        if (!g->grad_face_normals) return MM_ERR_NULL_POINTER;
        mm::clear_stale_error();
        return mm::launch_vertex_bwd(d, g, w, s);
    }
    if ((!g->grad_rgba && !d->fused_gt) || !g->grad_textures || !g->grad_lights) return MM_ERR_NULL_POINTER;
    if (d->no_mask && !g->grad_bg) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    st = mm::launch_raster_bwd(d, g, w, s);
    if (st != MM_OK) return st;
    return mm::launch_vertex_bwd(d, g, w, s);
}

static int check_recon(const MMReconDesc* d, bool backward) {
    if (!d) return MM_ERR_NULL_POINTER;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0) return MM_ERR_BAD_SHAPE;
    if (!d->pred || !d->gt) return MM_ERR_NULL_POINTER;
    if (!backward && !d->loss) return MM_ERR_NULL_POINTER;
    if (backward && !d->grad_pred) return MM_ERR_NULL_POINTER;
    if (d->contour > 0.f && (d->H < 4 || d->W < 4)) return MM_ERR_BAD_SHAPE;
    if (!d->workspace || d->workspace_bytes < mm_recon_query_workspace(d)) return MM_ERR_WORKSPACE;
    return MM_OK;
}

size_t mm_recon_query_workspace(const MMReconDesc* d) {
    if (!d || d->B <= 0) return 0;
    return mm::recon_workspace_bytes(d);
}

int mm_recon_data_forward(const MMReconDesc* d, mm_stream_t stream) {
    int st = check_recon(d, false);
    if (st != MM_OK) return st;
    mm::clear_stale_error();
    return mm::launch_recon_fwd(d, (hipStream_t)stream);
}

int mm_recon_data_backward(const MMReconDesc* d, mm_stream_t stream) {
    int st = check_recon(d, true);
    if (st != MM_OK) return st;
    mm::clear_stale_error();
    return mm::launch_recon_bwd(d, (hipStream_t)stream);
}

const float* mm_recon_data_totals(const MMReconDesc* d) {
    if (!d || d->B <= 0 || !d->workspace || d->workspace_bytes < mm_recon_query_workspace(d)) return nullptr;
    return mm::recon_totals(d);
}

int mm_nearest_neighbour(int32_t B, int32_t N, int32_t M, const float* x, const float* y, float* dist, int32_t* idx,
                         mm_stream_t stream) {
    if (!x || !y || !dist || !idx) return MM_ERR_NULL_POINTER;
    if (B <= 0 || N <= 0 || M <= 0) return MM_ERR_BAD_SHAPE;
    mm::clear_stale_error();
    return mm::launch_nn(B, N, M, x, y, dist, idx, (hipStream_t)stream);
}

int mm_chamfer_nearest(int32_t B, int32_t N, int32_t M, const float* x, const float* y, float* dist_x, int32_t* idx_x,
                       float* dist_y, int32_t* idx_y, mm_stream_t stream) {
    if (!x || !y || !dist_x || !idx_x || !dist_y || !idx_y) return MM_ERR_NULL_POINTER;
    if (B <= 0 || N <= 0 || M <= 0) return MM_ERR_BAD_SHAPE;
    if (B > 65535) return MM_ERR_UNSUPPORTED;
    mm::clear_stale_error();
    return mm::launch_nn_both(B, N, M, x, y, dist_x, idx_x, dist_y, idx_y, (hipStream_t)stream);
}

static int check_reg(const MMMeshRegDesc* d, bool backward) {
    if (!d) return MM_ERR_NULL_POINTER;
    if (d->B <= 0 || d->V <= 0 || d->F <= 0 || d->E < 0) return MM_ERR_BAD_SHAPE;
    if (d->terms == 0 || d->terms >= (1u << MM_REG_TERMS)) return MM_ERR_BAD_SHAPE;
    const unsigned need_v = (1u << MM_REG_EDGE) | (1u << MM_REG_DEPTH) | (1u << MM_REG_DEPTHR) | (1u << MM_REG_DEPTHC);
    const unsigned need_d = (1u << MM_REG_LAPLACIAN) | (1u << MM_REG_DEFORM) | (1u << MM_REG_FLIP);
    if ((d->terms & need_v) && (!d->vertices || !d->sign_init)) return MM_ERR_NULL_POINTER;
    if ((d->terms & need_d) && !d->delta_vertices) return MM_ERR_NULL_POINTER;
    if ((d->terms & (1u << MM_REG_FLAT)) && (!d->face_normals || !d->edge2faces || (backward && (!d->fe_offsets || !d->fe_items)))) return MM_ERR_NULL_POINTER;
    if ((d->terms & (1u << MM_REG_LAPLACIAN)) && (!d->lap_offsets || !d->lap_cols || !d->lap_vals ||
                                                   (backward && (!d->lapT_offsets || !d->lapT_cols || !d->lapT_vals)))) return MM_ERR_NULL_POINTER;
    if ((d->terms & (1u << MM_REG_EDGE)) && (!d->edges || (backward && (!d->ve_offsets || !d->ve_items)))) return MM_ERR_NULL_POINTER;
    if ((d->terms & (1u << MM_REG_FLIP)) && (!d->flip_index || !d->sign_init || (backward && (!d->flipT_offsets || !d->flipT_items)))) return MM_ERR_NULL_POINTER;
    if ((d->terms & ((1u << MM_REG_DEPTHR) | (1u << MM_REG_DEPTHC))) && !(d->ratio > 0.f)) return MM_ERR_BAD_SHAPE;
    if (!backward && !d->losses) return MM_ERR_NULL_POINTER;
    if (!d->workspace || d->workspace_bytes < mm_mesh_reg_query_workspace(d)) return MM_ERR_WORKSPACE;
    return MM_OK;
}

size_t mm_mesh_reg_query_workspace(const MMMeshRegDesc* d) {
    if (!d || d->B <= 0 || d->V <= 0 || d->E < 0) return 0;
    return mm::reg_workspace_bytes(d);
}

int mm_mesh_reg_forward(const MMMeshRegDesc* d, mm_stream_t stream) {
    const int st = check_reg(d, false);
    if (st != MM_OK) return st;
    mm::clear_stale_error();
    return mm::launch_reg_fwd(d, (hipStream_t)stream);
}

int mm_mesh_reg_backward(const MMMeshRegDesc* d, const MMMeshRegGrads* g, mm_stream_t stream) {
    const int st = check_reg(d, true);
    if (st != MM_OK) return st;
    if (!g || !g->weights) return MM_ERR_NULL_POINTER;
    if (!g->grad_vertices && !g->grad_delta_vertices && !g->grad_face_normals) return MM_ERR_NULL_POINTER;
    if ((g->grad_vertices && (!d->vertices || !d->sign_init)) || (g->grad_delta_vertices && !d->delta_vertices) ||
        (g->grad_face_normals && !d->face_normals))
        return MM_ERR_NULL_POINTER;                              // a gradient is only defined for an input that was given
    mm::clear_stale_error();
    return mm::launch_reg_bwd(d, g, (hipStream_t)stream);
}

static int check_att(const MMAttLossDesc* d) {
    if (!d) return MM_ERR_NULL_POINTER;
    if (d->B <= 0 || d->V <= 0 || d->Ht <= 0 || d->Wt <= 0) return MM_ERR_BAD_SHAPE;
    const MMAttributes* two[2] = {&d->pred, &d->target};
    for (const MMAttributes* m : two)
        if (!m->azimuths || !m->elevations || !m->distances || !m->biases || !m->vertices || !m->textures || !m->lights) return MM_ERR_NULL_POINTER;
    if (!d->workspace || d->workspace_bytes < mm_attribute_loss_query_workspace(d)) return MM_ERR_WORKSPACE;
    return MM_OK;
}

size_t mm_attribute_loss_query_workspace(const MMAttLossDesc* d) {
    if (!d || d->B <= 0 || d->Ht <= 0 || d->Wt <= 0) return 0;
    return mm::att_workspace_bytes(d);
}

int mm_attribute_loss_forward(const MMAttLossDesc* d, mm_stream_t stream) {
    const int st = check_att(d);
    if (st != MM_OK) return st;
    if (!d->losses) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    return mm::launch_att_fwd(d, (hipStream_t)stream);
}

int mm_attribute_loss_backward(const MMAttLossDesc* d, const MMAttLossGrads* g, mm_stream_t stream) {
    const int st = check_att(d);
    if (st != MM_OK) return st;
    if (!g || !g->weights) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    return mm::launch_att_bwd(d, g, (hipStream_t)stream);
}

static int check_texflow(const MMTexFlowDesc* d) {
    if (!d) return MM_ERR_NULL_POINTER;
    if (d->B <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0) return MM_ERR_BAD_SHAPE;
    if (d->B > 65535 || (d->Ho + 3) / 4 > 65535) return MM_ERR_UNSUPPORTED;     // grid y / z limits
    if (!d->image || !d->flow) return MM_ERR_NULL_POINTER;
    return MM_OK;
}

int mm_texture_flow_forward(const MMTexFlowDesc* d, mm_stream_t stream) {
    const int st = check_texflow(d);
    if (st != MM_OK) return st;
    if (!d->textures) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    return mm::launch_texflow_fwd(d, (hipStream_t)stream);
}

int mm_texture_flow_backward(const MMTexFlowDesc* d, const MMTexFlowGrads* g, mm_stream_t stream) {
    const int st = check_texflow(d);
    if (st != MM_OK) return st;
    if (!g || !g->grad_textures || !g->grad_flow) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    return mm::launch_texflow_bwd(d, g, (hipStream_t)stream);
}

int mm_build_vertex_corner_csr(int32_t V, int32_t F, const int32_t* faces, int32_t* offsets, int32_t* items) {
    if (!faces || !offsets || !items) return MM_ERR_NULL_POINTER;
    if (V <= 0 || F <= 0) return MM_ERR_BAD_SHAPE;
    for (int i = 0; i <= V; ++i) offsets[i] = 0;
    for (int i = 0; i < 3 * F; ++i) {
        if (faces[i] < 0 || faces[i] >= V) return MM_ERR_BAD_SHAPE;
        ++offsets[faces[i] + 1];
    }
    for (int i = 0; i < V; ++i) offsets[i + 1] += offsets[i];
    // counting sort: corners visited ascending, so each vertex's list is ascending; offsets doubles as the cursor
    for (int i = 0; i < 3 * F; ++i) items[offsets[faces[i]]++] = i;
    for (int v = V; v > 0; --v) offsets[v] = offsets[v - 1];
    offsets[0] = 0;
    return MM_OK;
}

int mm_build_vertex_corner_table(int32_t V, int32_t F, const int32_t* faces, int32_t stride, int32_t* table) {
    if (!faces) return MM_ERR_NULL_POINTER;
    if (V <= 0 || F <= 0) return MM_ERR_BAD_SHAPE;
    int32_t* cnt = (int32_t*)calloc((size_t)V, sizeof(int32_t));                 // (a host helper: the GPU path never allocates)
    if (!cnt) return MM_ERR_WORKSPACE;
    int32_t valence = 0;
    for (int i = 0; i < 3 * F; ++i) {
        if (faces[i] < 0 || faces[i] >= V) { free(cnt); return MM_ERR_BAD_SHAPE; }
        if (++cnt[faces[i]] > valence) valence = cnt[faces[i]];
    }
    if (!table) { free(cnt); return valence; }
    if (stride < valence) { free(cnt); return MM_ERR_BAD_SHAPE; }
    for (size_t i = 0; i < (size_t)V * stride * 4; ++i) table[i] = -1;
    for (int v = 0; v < V; ++v) cnt[v] = 0;
    for (int i = 0; i < 3 * F; ++i) {                             // corners visited ascending: every vertex's list is ascending
        const int v = faces[i], f = i / 3;
        int32_t* e = table + ((size_t)v * stride + cnt[v]++) * 4;
        e[0] = i; e[1] = faces[f * 3]; e[2] = faces[f * 3 + 1]; e[3] = faces[f * 3 + 2];
    }
    free(cnt);
    return MM_OK;
}

const char* mm_status_string(int status) {
    switch (status) {
        case MM_OK: return "ok";
        case MM_ERR_NULL_POINTER: return "required pointer is NULL";
        case MM_ERR_BAD_SHAPE: return "bad or inconsistent size";
        case MM_ERR_WORKSPACE: return "workspace missing, misaligned or too small";
        case MM_ERR_LAUNCH: return "HIP launch failed";
        case MM_ERR_UNSUPPORTED: return "unsupported option";
        default: return "unknown status";
    }
}

const char* mm_last_error_detail(void) {
    static thread_local char buf[160];
    const mm::LaunchError& e = mm::last_launch_error();
    if (e.code == hipSuccess) return "";
    snprintf(buf, sizeof buf, "%s: %s (hipError %d)", e.what, hipGetErrorString(e.code), (int)e.code);
    return buf;
}

size_t mm_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(MMRenderDesc);    case 1: return sizeof(MMRenderGrads);   case 2: return sizeof(MMReconDesc);
        case 3: return sizeof(MMMeshRegDesc);   case 4: return sizeof(MMMeshRegGrads);  case 5: return sizeof(MMAttLossDesc);
        case 6: return sizeof(MMAttLossGrads);  case 7: return sizeof(MMTexFlowDesc);   case 8: return sizeof(MMTexFlowGrads);
        case 9: return sizeof(MMPrepareDesc);   case 10: return sizeof(MMPrepareGrads); case 11: return sizeof(MMDibrDesc);
        case 12: return sizeof(MMDibrGrads);    case 13: return sizeof(MMTexMapDesc);   case 14: return sizeof(MMTexMapGrads);
        case 15: return sizeof(MMShDesc);       case 16: return sizeof(MMShGrads);      case 17: return sizeof(MMMaskIouDesc);
        default: return 0;
    }
}

int mm_abi_version(void) { return MM_ABI_VERSION; }

}  // extern "C"
