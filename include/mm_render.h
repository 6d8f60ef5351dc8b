/*
 * mm_render.h -- C ABI of libmm_render.so: the MI355X (gfx950) differentiable render + reconstruction-loss path of
 * 3D-Magic-Mirror.
 *
 * This is the drop-in boundary for the path
 *     DiffRender.render        /root/reference/networks.py:258-324
 *     DiffRender.recon_data    /root/reference/networks.py:364-390
 * and, below it, for what the reference reaches through kaolin (NVIDIAGameWorks/kaolin v0.12.0, not vendored):
 *     kaolin.render.mesh.prepare_vertices / dibr_rasterization / texture_mapping / spherical_harmonic_lighting
 *     (call sites networks.py:284-306) -> kaolin._C.render.mesh.{packed_rasterize_forward_cuda,
 *     rasterize_backward_cuda, dibr_soft_mask_forward_cuda, dibr_soft_mask_backward_cuda},
 *     kaolin.metrics.render.mask_iou (call site networks.py:377).
 *
 * Rules of the ABI
 *   - plain C: raw DEVICE pointers, sizes and scalars only; no torch / C++ types.
 *   - every function returns MM_OK (0) or a negative MMStatus; nothing throws across the boundary.
 *   - the library never allocates: the caller owns every buffer including the workspace
 *     (size from mm_query_workspace / mm_recon_query_workspace) and keeps the render workspace alive, unmodified,
 *     between mm_render_forward and the matching mm_render_backward.
 *   - all work is enqueued on the given HIP stream (pass the hipStream_t as a void*; NULL = the null stream);
 *     no host synchronisation, no global state, re-entrant.
 *   - all floating point is fp32; indices are int32.  Tensors are dense row-major unless strides are given.
 */
#ifndef MM_RENDER_H
#define MM_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mm_stream_t; /* hipStream_t */

typedef enum MMStatus {
    MM_OK = 0,
    MM_ERR_NULL_POINTER = -1,   /* a required pointer is NULL */
    MM_ERR_BAD_SHAPE = -2,      /* a size is <= 0 or inconsistent */
    MM_ERR_WORKSPACE = -3,      /* workspace missing or smaller than mm_query_workspace() */
    MM_ERR_LAUNCH = -4,         /* hipLaunchKernel / hipMemsetAsync reported an error */
    MM_ERR_UNSUPPORTED = -5     /* outside what the kernels implement (knum <= 0, image side > 65535, > MM_DIBR_MAX_D channels) */
} MMStatus;

/* --------------------------------------------------------------------------------------------------------------------
 * Render: replaces DiffRender.render (networks.py:258-324), i.e. camera (smr_utils.py:257-311) -> prepare_vertices ->
 * dibr_rasterization -> texture_mapping -> spherical_harmonic_lighting -> composite -> clamp -> cat(soft mask).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct MMRenderDesc {
    /* sizes */
    int32_t B, H, W;            /* batch, image rows (= round(ratio*image_size)), image cols (= image_size) */
    int32_t V, F;               /* template vertices / faces */
    int32_t Ht, Wt;             /* texture rows / cols */
    int32_t no_mask;            /* 1: composite over bg then shade (trainer's --bg); 0: white background  (:307-313) */
    int32_t knum;               /* dibr_rasterization knum (30) */
    /* constants: cam_proj (networks.py:172-174) and the dibr_rasterization defaults */
    float proj[3];              /* [1/(ratio'*tan(fovy/2)), 1/tan(fovy/2), -1] */
    float sigmainv, boxlen, multiplier, eps; /* 7000, 0.02, 1000, 1e-8 */
    /* static template data (device) */
    const int32_t* faces;       /* (F,3) vertex ids */
    const float* face_uvs;      /* (F,3,2) raw OBJ uv of every corner (networks.py:196-202) */
    const int32_t* vc_table;    /* (V,vc_stride,4) vertex -> incident corners, fixed stride (mm_build_vertex_corner_table): entry = {face*3 + corner,
                                 * the face's three vertex ids}, ascending, padded with {-1,..}   (backward only; may be NULL forward) */
    int32_t vc_stride;          /* entries per vertex (>= the largest valence of the template) */
    /* per-sample attributes (device), the 'attributes' dict of networks.py:259-270 */
    const float* vertices;      /* (B,V,3) */
    const float* textures;      /* (B,3,Ht,Wt) */
    const float* lights;        /* (B,9) */
    const float* bg;            /* (B,3,H,W); required iff no_mask */
    const float* azimuths;      /* (B) degrees */
    const float* elevations;    /* (B) degrees */
    const float* distances;     /* (B) */
    const float* biases;        /* (B,2) */
    /* outputs (device) */
    float* rgba;                /* (B,H,W,4): NHWC storage; the reference returns the (B,4,H,W) permute VIEW of it (:317).
                                 * Written by mm_render_forward; NOT read by mm_render_backward (may be NULL there, fused or not) */
    int32_t* face_idx;          /* (B,H,W): winning face per pixel, -1 = none (kaolin returns int64; int32 here) */
    float* face_normals;        /* (B,F,3): unit face normals in camera space = attributes['face_normals'] (:319) */
    float* imnormal;            /* (B,H,W,3) or NULL: attributes['imnormal'] (:320, "visualize only") */
    /* scratch */
    void* workspace;            /* >= mm_query_workspace(desc) bytes, 256-byte aligned */
    size_t workspace_bytes;
    /* optional profiling: NULL, or an array of 2*MM_PROF_RENDER_SLOTS hipEvent_t created by the caller; the library
     * records events [2*slot] / [2*slot+1] on the stream immediately before / after the kernel of that slot. */
    void** prof_events;
    /* optional FUSED reconstruction loss = DiffRender.recon_data (networks.py:364-390; its contour term: fused_contour below) folded into the render
     * kernels: with fused_gt set, mm_render_forward also reduces the loss terms while it shades, and mm_render_backward
     * derives dL/d rgba on the fly from fused_gt and the prediction it re-forms per pixel, bit for bit (`rgba` is not read back: the caller
     * may already have overwritten it; MMRenderGrads.grad_rgba is then ignored and may be NULL) and writes the loss value.
     * Same arithmetic as mm_recon_data_forward/backward; saves three launches and the grad_rgba round trip. */
    const float* fused_gt;          /* (B,4,H,W) dense rgb + mask, or NULL */
    float fused_image_weight;       /* DiffRender.image_weight */
    float* fused_loss;              /* (1) device scalar, written by mm_render_backward; may be NULL */
    const float* fused_grad_loss;   /* (1) device scalar dL/dloss, or NULL for 1 */
    int32_t options;                /* bit set of MM_OPT_* (below); 0 = the semantics of SURVEY.md 8(a) */
    /* 1: GEOMETRY ONLY -- the call site that discards the image and keeps attributes['face_normals'] (trainer.py:367:
     * `_, Aire = diffRender.render(**Aire)`).  mm_render_forward then runs the vertex stage alone (camera, prepare_vertices,
     * face_normals; nothing is rasterised, rgba / face_idx / imnormal are not written and may be NULL); mm_render_backward takes
     * MMRenderGrads.grad_face_normals (required) and writes grad_vertices and the four camera gradients only -- the texture, light and
     * background gradients of such a render are identically zero and are NOT written (their pointers may be NULL). */
    int32_t geometry_only;
    /* optional, may be NULL: one int32 the device can write -- device memory, or PINNED HOST memory, which the host can then poll without
     * synchronising.  mm_render_backward adds to it the number of texture-gradient records it had to drop (see mm_query_workspace /
     * mm_render_status): 0 stays 0.  Lets a caller that never synchronises (autograd nodes, captured graphs) still turn an overflowing
     * record pool into an error one step later instead of training on NaN texture gradients. */
    int32_t* status_flag;
    /* the fused loss's contour weight (recon_data's `contour` argument, networks.py:379-388; trainer.py:441 passes opt.lambda_contour): 0 = no
     * contour term.  > 0 needs H % 4 == 0 and W % 4 == 0 (then F.interpolate's two nearest resamplings pick the top-left pixel of every
     * 4x4 block, which lies in the pixel's own 8x8 screen tile; other sizes: MM_ERR_BAD_SHAPE -- use mm_recon_data_* for those).  Ignored
     * without fused_gt. */
    float fused_contour;
    /* DEFERRED fusion, mm_render_backward only (ABI 6): the un-modified trainer's order of calls -- `render`, then `recon_data(pred, gt)` on the image
     * that render wrote (trainer.py:276,441) -- with the fused backward.  The loss VALUE was formed by mm_recon_data_forward on the image (its own
     * launches, its own bits); fused_totals = that call's per-image totals, mm_recon_data_totals(recon desc): (B,4) floats in ITS workspace, which the
     * caller keeps alive until this backward has run.  With fused_gt AND fused_totals set, mm_render_backward forms dL/d rgba per pixel with
     * mm_recon_data_backward's own expressions from those totals and the prediction it re-forms -- bit for bit the gradient mm_recon_data_backward
     * would have written -- without that launch and without the grad_rgba round trip.  MMRenderGrads.grad_rgba, if not NULL, is ADDED to it (the
     * image's other consumers).  fused_contour must be 0 (the contour term's gradient is formed in another order by the fused kernels: use
     * mm_recon_data_backward for it); fused_loss is not written; the forward of this render ran WITHOUT fused_gt.  NULL: off. */
    const float* fused_totals;
} MMRenderDesc;

/* MMRenderDesc.options / MMDibrDesc.options: 0 = the semantics of SURVEY.md 8(a) (the oracle's defaults).  The bits switch,
 * one by one, the choices that SURVEY.md Appendix C lists as recalled from kaolin's sources and not re-verifiable here
 * (kaolin is not vendored): a maintainer with a CUDA box and real kaolin can pin the path by flipping a bit instead of
 * editing kernels.  oracle/mm_oracle.inc takes the same bits, and tests/ hold HIP == oracle for every one of them. */
enum { MM_OPT_WALK_BLOCK = 1 << 1,         /* tuning: force the 256-thread / cooperative-heavy-tile shape of the walk kernels ...          */
       MM_OPT_WALK_WAVE = 1 << 2,          /* ... or the one-wave-per-tile shape (default: chosen by screen-bin size and batch).  Identical forward outputs;
                                            * identical gradients too, except that with screen bins larger than a tile the 256-thread shape sweeps a few
                                            * more faces over their inflated boxes: the same integer sums cut into other items, <= 1e-9 of a gradient's maximum */
       MM_OPT_CULL_STRICT = 1 << 4,        /* rasterise faces with face_normals_z > 0 instead of >= 0                    (App. C-1) */
       MM_OPT_SOFT_SKIP_CULLED = 1 << 5,   /* the soft mask skips the faces the colour pass culls                          (App. C-1) */
       MM_OPT_BBOX_HALF_OPEN = 1 << 6,     /* a pixel centre exactly on a face's bbox edge is outside (<= / >= reject)     (App. C-4) */
       MM_OPT_BARY_ONE_MINUS = 1 << 7,     /* barycentrics as w1 = k1/(S+eps), w2 = k2/(S+eps), w0 = 1 - w1 - w2 (eps added, not
                                            * copysign'd) instead of three edge functions / copysign-padded sum          (App. C-3) */
       MM_OPT_SH_ORDER_XYZ = 1 << 8,       /* SH linear bands in x,y,z order and quadratic bands xy,yz,3z^2-1,xz,x^2-y^2 paired with
                                            * lights 1..8 in THAT order (instead of x,z,y / xy,yz,z^2,xz,x^2-y^2)          (App. C-6) */
       MM_OPT_WALK_QUEUE = 1 << 10,        /* tuning: force the compacting-queue form of the forward walk (default: screen bins larger than a tile) ...           */
       MM_OPT_WALK_BATCH = 1 << 11,        /* ... or the per-batch form (default: 8-pixel bins); identical results                                              */
       MM_OPT_MANY_IN_FLIGHT = 1 << 12,    /* hint, identical results: the caller keeps several independent calls in flight (several streams), so every launch
                                            * shares the chip -- the kernels take the shapes large batches take on their own (forward walk: one tile per
                                            * workgroup for 8-pixel screen bins; face sweep of the backward: four lanes per item).  Four B=48 steps on four
                                            * streams: +3 % images/s; ONE step at a time: -14 % (profiles/r06_many_in_flight_ab.md)                       */
       MM_OPT_BBOX_MIN_CLOSED_MAX_OPEN = 1 << 9 };  /* bbox test [min, max): reject x < min || x >= max -- the third form upstream may have,
                                            * between the closed default and MM_OPT_BBOX_HALF_OPEN (which opens both borders)  (App. C-4) */

enum { MM_PROF_VERTEX_FWD = 0, MM_PROF_RASTER_FWD = 1, MM_PROF_PIXEL_BWD = 2, MM_PROF_GATHER_BWD = 3, MM_PROF_VERTEX_BWD = 4,
       MM_PROF_ORDER = 5, MM_PROF_RENDER_SLOTS = 6 };
enum { MM_PROF_RECON_PARTIAL = 0, MM_PROF_RECON_FINAL = 1, MM_PROF_RECON_BWD = 2, MM_PROF_RECON_CONTOUR = 3,
       MM_PROF_RECON_SLOTS = 4 };

/* Gradients of one render call.  Every non-NULL output is OVERWRITTEN (the library zero-fills what it accumulates). */
typedef struct MMRenderGrads {
    const float* grad_rgba;          /* (B,H,W,4) NHWC, dL/d rgba; required unless MMRenderDesc.fused_gt is set */
    const float* grad_face_normals;  /* (B,F,3) or NULL: dL/d attributes['face_normals'] (used by calc_reg_loss, :422-431) */
    float* grad_vertices;            /* (B,V,3) */
    float* grad_textures;            /* (B,3,Ht,Wt) */
    float* grad_lights;              /* (B,9) */
    float* grad_bg;                  /* (B,3,H,W); required iff no_mask */
    float* grad_azimuths;            /* (B) per degree */
    float* grad_elevations;          /* (B) per degree */
    float* grad_distances;           /* (B) */
    float* grad_biases;              /* (B,2) */
} MMRenderGrads;

/* The MINIMUM workspace for the shape.  Everything in it is sized for the worst case except the pool of texture-gradient records
 * (one per covered pixel and texture tile under its bilinear footprint), which holds 9/8 records per pixel: a fully covered image
 * with one footprint in eight across a tile border.  A workspace_bytes above the minimum is used: the excess enlarges the images'
 * record arrays (24 bytes per record and image).  An image that still runs out (texture coordinates that put most pixels on the
 * corners of 32x32-texel tiles) gets NaN in ALL of its grad_textures texels -- never a silently short sum -- and mm_render_status
 * says how many records were dropped. */
size_t mm_query_workspace(const MMRenderDesc* desc);
int mm_render_forward(const MMRenderDesc* desc, mm_stream_t stream);
int mm_render_backward(const MMRenderDesc* desc, const MMRenderGrads* grads, mm_stream_t stream);
/* After mm_render_backward and before the next mm_render_forward on the same workspace: copies the per-image counts of dropped
 * texture-gradient records to dropped_host (B ints, may be NULL) and returns MM_OK if all are zero, MM_ERR_WORKSPACE otherwise
 * (a NULL desc / workspace: MM_ERR_NULL_POINTER; a bad shape, a workspace below the minimum or misaligned: MM_ERR_BAD_SHAPE).
 * The one entry point that SYNCHRONISES the stream (a diagnostic, not part of a step). */
int mm_render_status(const MMRenderDesc* desc, mm_stream_t stream, int32_t* dropped_host);
/* Fused mode only (desc->fused_gt and desc->fused_loss set), after mm_render_forward: writes the recon_data value of the batch
 * (networks.py:364-390, contour = 0) to desc->fused_loss from the sums the forward left in the workspace -- for callers that need
 * the loss before they run the backward (the autograd API DiffRender.render_recon).  mm_render_backward writes the same value. */
int mm_render_fused_loss(const MMRenderDesc* desc, mm_stream_t stream);
/* Tools only (profiles/tools): byte offsets inside the render workspace of out[0] = chunkmap (B,F) int2, out[1] = sweep items (B,item_cap)
 * int2, out[2] = nitems (B) int2, out[3] = per-item partial sums (B,item_cap,12) float; out[4] = item_cap; out[5] = gp (B,H,W,2) float4,
 * out[6] = gp2 (B,H,W) float, out[7] = soft (B,H,W) float2, out[8] = per-texture-tile record counts of the last backward (B,ntiles) int
 * followed by the list offsets + 1 (B,ntiles) and the records dropped (B); out[9] = ntiles; out[10] = records an image's array holds;
 * out[11] = the forward's per-tile footprint counts (B,ntiles) int.  `out` has room for 16 values.  Returns 0, or MM_ERR_*. */
int mm_debug_workspace_layout(const MMRenderDesc* desc, size_t* out8);

/* --------------------------------------------------------------------------------------------------------------------
 * Reconstruction loss: replaces DiffRender.recon_data (networks.py:364-390) incl. kaolin mask_iou (:377) and the
 * optional contour term (:379-387):  loss = image_weight * mean|pred*gm+(1-gm) - (gt*gm+(1-gm))| + (1 - mean_b IoU_b)
 *                                           [+ contour * mean((c(pred_mask) - c(gt_mask))^2)].
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct MMReconDesc {
    int32_t B, H, W;
    const float* pred;           /* rgba prediction, element strides below (NHWC storage from mm_render_forward: {4HW,1,4W,4}) */
    int64_t pred_strides[4];     /* strides of (b, channel, y, x) in elements */
    const float* gt;             /* (B,4,H,W) dense: rgb + binary mask */
    float image_weight;          /* DiffRender.image_weight */
    float contour;               /* lambda_contour; <= 0 disables the term */
    float* loss;                 /* (1) device scalar, overwritten */
    /* backward only */
    const float* grad_loss;      /* (1) device scalar dL/dloss, or NULL for 1 */
    float* grad_pred;            /* same strides as pred; overwritten */
    void* workspace;             /* >= mm_recon_query_workspace(desc) bytes; forward fills it, backward reads it */
    size_t workspace_bytes;
    void** prof_events;          /* optional: 2*MM_PROF_RECON_SLOTS hipEvent_t, as in MMRenderDesc */
} MMReconDesc;

size_t mm_recon_query_workspace(const MMReconDesc* desc);
int mm_recon_data_forward(const MMReconDesc* desc, mm_stream_t stream);
int mm_recon_data_backward(const MMReconDesc* desc, mm_stream_t stream);
/* Where mm_recon_data_forward left the per-image totals {sum|pi-gi|, sum p*g, sum p+g-p*g, contour sum} (B,4) inside desc->workspace: what
 * MMRenderDesc.fused_totals takes (deferred fusion).  NULL if desc or its workspace is NULL / too small. */
const float* mm_recon_data_totals(const MMReconDesc* desc);

/* --------------------------------------------------------------------------------------------------------------------
 * Nearest neighbour of every point of x (B,N,3) in y (B,M,3): squared distance (B,N) and index (B,N) int32, lowest index
 * on ties.  The O(N*M) half of pytorch3d.loss.chamfer_distance (knn_points, K=1) that DiffRender.recon_att(chamfer=True)
 * needs (networks.py:342,356); the differentiable tail is a gather.
 * ------------------------------------------------------------------------------------------------------------------ */
int mm_nearest_neighbour(int32_t B, int32_t N, int32_t M, const float* x, const float* y, float* dist, int32_t* idx,
                         mm_stream_t stream);
/* Both directions of pytorch3d.loss.chamfer_distance in ONE launch (networks.py:342,356 always needs both): for every x its
 * nearest y (dist_x, idx_x: (B,N)) and for every y its nearest x (dist_y, idx_y: (B,M)).  Same results as two calls above. */
int mm_chamfer_nearest(int32_t B, int32_t N, int32_t M, const float* x, const float* y, float* dist_x, int32_t* idx_x,
                       float* dist_y, int32_t* idx_y, mm_stream_t stream);

/* --------------------------------------------------------------------------------------------------------------------
 * Mesh regularisers: replaces DiffRender.calc_reg_loss / calc_reg_edge / calc_reg_depth /
 * calc_reg_depthR / calc_reg_depthC / calc_reg_deform / recon_flip(L1=False) (networks.py:392-491; orchestrated by
 * trainer.py:54-74), one launch per direction for any subset of the terms.  losses[k] is the reference's value of term k
 * (calc_reg_loss = lambda_lpl * losses[LAPLACIAN] + lambda_flat * losses[FLAT]; calc_reg_edge = losses[EDGE], which already
 * carries the reference's 0.1); terms that were not requested read 0.
 * ------------------------------------------------------------------------------------------------------------------ */
enum { MM_REG_LAPLACIAN = 0, MM_REG_FLAT = 1, MM_REG_EDGE = 2, MM_REG_DEPTH = 3, MM_REG_DEPTHR = 4, MM_REG_DEPTHC = 5,
       MM_REG_DEFORM = 6, MM_REG_FLIP = 7, MM_REG_TERMS = 8 };

typedef struct MMMeshRegDesc {
    int32_t B, V, F, E;         /* batch, template vertices / faces / unique edges */
    uint32_t terms;             /* bit k set: compute term k */
    /* static template tables (device).  The (V,V) laplacian (networks.py:249) and its transpose as CSR, diagonal included */
    const int32_t* lap_offsets;   const int32_t* lap_cols;   const float* lap_vals;
    const int32_t* lapT_offsets;  const int32_t* lapT_cols;  const float* lapT_vals;    /* backward only */
    const int32_t* edges;         /* (E,2)  vertex ids                                   (:220-233) */
    const int32_t* edge2faces;    /* (E,2)  the two faces of every edge                  (:235-246) */
    const int32_t* ve_offsets;    const int32_t* ve_items;   /* vertex -> edge*2 + end   (backward only) */
    const int32_t* fe_offsets;    const int32_t* fe_items;   /* face   -> edge*2 + side  (backward only) */
    const int32_t* flip_index;    /* (V)    mirrored partner of every vertex             (:215-217) */
    const int32_t* flipT_offsets; const int32_t* flipT_items; /* u -> {v : flip_index[v] == u}  (backward only) */
    const float* sign_init;       /* (V)    sign of the template's z                     (:213) */
    /* inputs (device); one may be NULL if no requested term reads it */
    const float* vertices;        /* (B,V,3)  EDGE, DEPTH, DEPTHR, DEPTHC */
    const float* delta_vertices;  /* (B,V,3)  LAPLACIAN, DEFORM, FLIP */
    const float* face_normals;    /* (B,F,3)  FLAT */
    float ratio, temp, eps;       /* DiffRender.ratio; calc_reg_depthR's temp (2); depthR / depthC eps (0.001) */
    float* losses;                /* (MM_REG_TERMS) device */
    void* workspace;              /* >= mm_mesh_reg_query_workspace bytes, 256-byte aligned, ZERO-FILLED by the caller before its
                                   * first use; the library leaves it ready for the next call.  The backward reads what the
                                   * forward of the same inputs left in it. */
    size_t workspace_bytes;
} MMMeshRegDesc;

typedef struct MMMeshRegGrads {
    const float* weights;         /* (MM_REG_TERMS) device: dL/d losses[k] */
    float* grad_vertices;         /* (B,V,3) or NULL; overwritten */
    float* grad_delta_vertices;   /* (B,V,3) or NULL; overwritten */
    float* grad_face_normals;     /* (B,F,3) or NULL; overwritten */
} MMMeshRegGrads;

size_t mm_mesh_reg_query_workspace(const MMMeshRegDesc* desc);
int mm_mesh_reg_forward(const MMMeshRegDesc* desc, mm_stream_t stream);
int mm_mesh_reg_backward(const MMMeshRegDesc* desc, const MMMeshRegGrads* grads, mm_stream_t stream);

/* --------------------------------------------------------------------------------------------------------------------
 * Attribute-reconstruction losses: replaces the seven means of DiffRender.recon_att
 * (networks.py:326-362; the chamfer variant of the shape term goes through mm_nearest_neighbour instead).
 * losses = { azim, elev, dist, bias, shape, texture, light }: mean |a-b| (l1 = 1) or mean (a-b)^2 (l1 = 0) over all
 * elements, azimuths / elevations through angle2xy (cos, sin of the angle in degrees).  The reference composes
 * loss_cam = azim * losses[0] + losses[1] + losses[2], loss_light = 0.1 * losses[6].
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct MMAttributes {     /* device pointers; in MMAttLossGrads any may be NULL (= that gradient is not wanted) */
    float* azimuths;              /* (B) degrees */
    float* elevations;            /* (B) degrees */
    float* distances;             /* (B) */
    float* biases;                /* (B,2) */
    float* vertices;              /* (B,V,3) */
    float* textures;              /* (B,3,Ht,Wt) */
    float* lights;                /* (B,9) */
} MMAttributes;

typedef struct MMAttLossDesc {
    int32_t B, V, Ht, Wt;
    int32_t l1;                   /* 1: mean |a-b| (opt.L1), 0: mean (a-b)^2 */
    MMAttributes pred, target;    /* read only; all seven required */
    float* losses;                /* (7) device */
    void* workspace;              /* >= mm_attribute_loss_query_workspace bytes, 256-byte aligned, ZERO-FILLED before its first use */
    size_t workspace_bytes;
} MMAttLossDesc;

typedef struct MMAttLossGrads {
    const float* weights;         /* (7) device: dL/d losses[k] */
    MMAttributes pred, target;    /* outputs, overwritten; NULL members are skipped */
} MMAttLossGrads;

size_t mm_attribute_loss_query_workspace(const MMAttLossDesc* desc);
int mm_attribute_loss_forward(const MMAttLossDesc* desc, mm_stream_t stream);
int mm_attribute_loss_backward(const MMAttLossDesc* desc, const MMAttLossGrads* grads, mm_stream_t stream);

/* --------------------------------------------------------------------------------------------------------------------
 * Texture-flow sampling: the tail of TextureEncoder.forward (network/model_res.py:597-612, makeup = 0),
 * i.e. the step that produces the texture the render path consumes:
 *   textures = cat([t, t.flip(2)], 2),  t = F.grid_sample(image, flow.permute(0,2,3,1), mode='bicubic', align_corners=True)
 * (zeros padding, ATen's bicubic: A = -0.75).  flow is taken channel-first, exactly as the decoder emits it.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct MMTexFlowDesc {
    int32_t B, C, H, W;         /* image batch, channels (3), rows, cols */
    int32_t Ho, Wo;             /* flow / sampled rows, cols; the texture has 2*Ho rows */
    const float* image;         /* (B,C,H,W) */
    const float* flow;          /* (B,2,Ho,Wo): x then y, in [-1,1] */
    float* textures;            /* (B,C,2*Ho,Wo); unused by the backward */
} MMTexFlowDesc;

typedef struct MMTexFlowGrads {
    const float* grad_textures; /* (B,C,2*Ho,Wo) */
    float* grad_flow;           /* (B,2,Ho,Wo); overwritten */
    float* grad_image;          /* (B,C,H,W) or NULL; overwritten (zero-filled on the stream, then float atomics) */
} MMTexFlowGrads;

int mm_texture_flow_forward(const MMTexFlowDesc* desc, mm_stream_t stream);
int mm_texture_flow_backward(const MMTexFlowDesc* desc, const MMTexFlowGrads* grads, mm_stream_t stream);

/* --------------------------------------------------------------------------------------------------------------------
 * Op boundary: the kaolin / pytorch3d operators the reference imports at module top (networks.py:6-19, trainer.py:31-40),
 * un-fused, one entry point per operator and direction.  The Python package 3d-magic-mirror_amd/shim exposes them under
 * kaolin's own module paths and signatures, so networks.py / trainer.py import and run unmodified.
 * They run the SAME device code as the fused render path (the same candidate walk, barycentrics, bilinear fetch and SH bands),
 * so face_idx is identical between the two boundaries.  Upstream semantics: NVIDIAGameWorks/kaolin v0.12.0 (not vendored;
 * restated in SURVEY.md 8(a)); gradients SURVEY.md Appendix A.
 * ------------------------------------------------------------------------------------------------------------------ */

/* kaolin.render.mesh.prepare_vertices(vertices, faces, camera_proj, camera_transform=...)   (call site networks.py:284-287):
 * vc = [v,1] @ T; vi = (vc.xy * proj.xy) / (vc.z * proj.z); gather by faces; unit normals with +1e-10 on the length. */
typedef struct MMPrepareDesc {
    int32_t B, V, F;
    float proj[3];                  /* camera_proj (3,1) */
    const int32_t* faces;           /* (F,3) */
    const int32_t* vc_offsets;      /* (V+1) vertex -> corner CSR (mm_build_vertex_corner_csr); backward only */
    const int32_t* vc_items;        /* (3F) */
    const float* vertices;          /* (B,V,3) */
    const float* transform;         /* (B,4,3) camera_transform [R;t] */
    float* face_vertices_camera;    /* (B,F,3,3) */
    float* face_vertices_image;     /* (B,F,3,2) */
    float* face_normals;            /* (B,F,3)   */
    void* workspace;                /* backward only: >= mm_prepare_vertices_query_workspace bytes (per-workgroup dT partials) */
    size_t workspace_bytes;
    const float* proj_device;       /* optional: camera_proj as 3 floats in DEVICE memory, read instead of proj[] -- a caller whose projection
                                     * is a device tensor (kaolin's prepare_vertices takes one) needs no device -> host read */
} MMPrepareDesc;

typedef struct MMPrepareGrads {
    const float* grad_face_vertices_camera;  /* (B,F,3,3) or NULL */
    const float* grad_face_vertices_image;   /* (B,F,3,2) or NULL */
    const float* grad_face_normals;          /* (B,F,3)   or NULL */
    float* grad_vertices;                    /* (B,V,3) overwritten */
    float* grad_transform;                   /* (B,4,3) overwritten, or NULL */
} MMPrepareGrads;

size_t mm_prepare_vertices_query_workspace(const MMPrepareDesc* desc);
int mm_prepare_vertices_forward(const MMPrepareDesc* desc, mm_stream_t stream);
int mm_prepare_vertices_backward(const MMPrepareDesc* desc, const MMPrepareGrads* grads, mm_stream_t stream);

/* kaolin.ops.mesh.face_normals(face_vertices (n,3,3), unit)   (call site networks.py:289): cross(v1-v0, v2-v0), optionally
 * divided by (length + 1e-10).  n = B*F faces.  backward: grad_normals (n,3) -> grad_face_vertices (n,3,3), overwritten. */
int mm_face_normals_forward(int64_t n, int32_t unit, const float* face_vertices, float* normals, mm_stream_t stream);
int mm_face_normals_backward(int64_t n, int32_t unit, const float* face_vertices, const float* grad_normals,
                             float* grad_face_vertices, mm_stream_t stream);

/* kaolin.render.mesh.dibr_rasterization(height, width, face_vertices_z, face_vertices_image, face_features, face_normals_z,
 * sigmainv=7000, boxlen=0.02, knum=30, multiplier=1000, eps=1e-8)   (call site networks.py:297-299)
 *   = rasterize (kaolin._C packed_rasterize_forward_cuda / rasterize_backward_cuda, K1/K2)
 *   + dibr_soft_mask (dibr_soft_mask_forward_cuda / _backward_cuda, K3/K4).
 * Outputs as kaolin returns them: interpolated features (zeros where uncovered), soft mask, int64 face_idx (-1 = none).
 * Gradients only to face_vertices_image and face_features (none to z / normals_z, like upstream). */
#define MM_DIBR_MAX_D 32            /* feature channels per corner the backward's LDS accumulators hold */
typedef struct MMDibrDesc {
    int32_t B, H, W, F, D, knum;
    float sigmainv, boxlen, multiplier, eps;
    const float* face_vertices_z;       /* (B,F,3) */
    const float* face_vertices_image;   /* (B,F,3,2) */
    const float* face_features;         /* (B,F,3,D) (a list of feature tensors is concatenated by the caller) */
    const float* face_normals_z;        /* (B,F) : faces with normal z >= 0 are rasterised */
    float* interpolated_features;       /* (B,H,W,D) */
    float* soft_mask;                   /* (B,H,W) */
    int64_t* face_idx;                  /* (B,H,W) */
    void* workspace;                    /* >= mm_dibr_query_workspace bytes, 256-byte aligned; filled by the forward, read by the backward */
    size_t workspace_bytes;
    int32_t options;                    /* MM_OPT_* bits (0 = defaults) */
} MMDibrDesc;

typedef struct MMDibrGrads {
    const float* grad_interpolated_features;   /* (B,H,W,D) or NULL */
    const float* grad_soft_mask;               /* (B,H,W)   or NULL */
    float* grad_face_vertices_image;           /* (B,F,3,2) overwritten */
    float* grad_face_features;                 /* (B,F,3,D) overwritten, or NULL */
} MMDibrGrads;

size_t mm_dibr_query_workspace(const MMDibrDesc* desc);
int mm_dibr_rasterization_forward(const MMDibrDesc* desc, mm_stream_t stream);
int mm_dibr_rasterization_backward(const MMDibrDesc* desc, const MMDibrGrads* grads, mm_stream_t stream);

/* kaolin.render.mesh.texture_mapping(texture_coordinates, texture_maps, mode)   (call site networks.py:305)
 * = F.grid_sample(maps, (2u-1, -(2v-1)), mode, align_corners=False, padding_mode='border').  N = points per batch item (H*W). */
enum { MM_TEXMAP_NEAREST = 0, MM_TEXMAP_BILINEAR = 1 };
typedef struct MMTexMapDesc {
    int32_t B, N, C, Ht, Wt, mode;
    const float* uv;                /* (B,N,2) */
    const float* textures;          /* (B,C,Ht,Wt) */
    float* out;                     /* (B,N,C) */
} MMTexMapDesc;
typedef struct MMTexMapGrads {
    const float* grad_out;          /* (B,N,C) */
    float* grad_uv;                 /* (B,N,2) overwritten, or NULL (zero for MM_TEXMAP_NEAREST) */
    float* grad_textures;           /* (B,C,Ht,Wt) overwritten, or NULL */
    /* optional scratch of >= mm_texture_mapping_backward_query_workspace bytes (256-byte aligned).  With it the texture gradient is
     * accumulated in 64-bit FIXED POINT (per-image power-of-two scale from max |grad_out|; integer adds commute) and is bitwise
     * reproducible; without it (NULL) the scatter uses float atomics (order-dependent in the last bits).
     * NON-FINITE grad_out: the two forms differ.  The fixed-point form scales by the image's max |grad_out|, so ONE NaN / inf element turns the
     * image's WHOLE texture gradient into NaN (loud, never a silently wrong finite value); the float-atomic form -- like ATen's grid_sampler and
     * kaolin -- poisons only the texels under that element's footprint.  Code that masks NaNs per texel must pass workspace = NULL. */
    void* workspace;
    size_t workspace_bytes;
} MMTexMapGrads;
size_t mm_texture_mapping_backward_query_workspace(const MMTexMapDesc* desc);
int mm_texture_mapping_forward(const MMTexMapDesc* desc, mm_stream_t stream);
int mm_texture_mapping_backward(const MMTexMapDesc* desc, const MMTexMapGrads* grads, mm_stream_t stream);

/* kaolin.render.mesh.spherical_harmonic_lighting(imnormal (B,N,3), lights (B,9)) -> (B,N)   (call site networks.py:306) */
typedef struct MMShDesc {
    int32_t B, N;
    const float* normals;           /* (B,N,3) */
    const float* lights;            /* (B,9) */
    float* out;                     /* (B,N) */
} MMShDesc;
typedef struct MMShGrads {
    const float* grad_out;          /* (B,N) */
    float* grad_normals;            /* (B,N,3) overwritten, or NULL */
    float* grad_lights;             /* (B,9) overwritten (zero-filled on the stream, one float atomic per wave and band), or NULL */
} MMShGrads;
int mm_sh_lighting_forward(const MMShDesc* desc, mm_stream_t stream);
int mm_sh_lighting_backward(const MMShDesc* desc, const MMShGrads* grads, mm_stream_t stream);

/* kaolin.metrics.render.mask_iou(lhs (B,H,W), rhs (B,H,W)) = 1 - mean_b[ sum(l*r) / (sum(l+r-l*r) + 1e-10) ]
 * (call sites networks.py:377, trainer.py:793,933).  sums: (B,2) device scratch written by the forward and read by the backward. */
typedef struct MMMaskIouDesc {
    int32_t B, N;                   /* N = H*W */
    const float* lhs; const float* rhs;
    float* sums;                    /* (B,2): {sum l*r, sum l+r-l*r} */
    float* loss;                    /* (1) */
} MMMaskIouDesc;
int mm_mask_iou_forward(const MMMaskIouDesc* desc, mm_stream_t stream);
int mm_mask_iou_backward(const MMMaskIouDesc* desc, const float* grad_loss, float* grad_lhs, float* grad_rhs, mm_stream_t stream);

/* --------------------------------------------------------------------------------------------------------------------
 * Host helpers (no GPU involved)
 * ------------------------------------------------------------------------------------------------------------------ */
/* Build the vertex -> corner CSR from HOST faces (F,3).  offsets: (V+1), items: (3F).  Returns MM_OK or an error. */
int mm_build_vertex_corner_csr(int32_t V, int32_t F, const int32_t* faces_host, int32_t* offsets_host, int32_t* items_host);
/* The same CSR built ON THE DEVICE from device-resident faces, enqueued on the stream (one small launch, no host round trip): what the
 * kaolin-shaped prepare_vertices needs when `faces` arrives as a fresh device tensor on every call (networks.py:272 re-uploads it).
 * offsets_dev (V+1), items_dev (3F), every vertex's list ascending like the host builder's.  Vertex ids outside [0, V) are counted into
 * *status_flag (optional; device or pinned host memory, added to) and left out of the lists.  V <= 12288. */
int mm_build_vertex_corner_csr_device(int32_t V, int32_t F, const int32_t* faces_dev, int32_t* offsets_dev, int32_t* items_dev,
                                      int32_t* status_flag, mm_stream_t stream);
/* The same adjacency with a fixed stride, as MMRenderDesc.vc_table wants it (one trip to memory for a vertex's corners AND their faces'
 * vertex ids, where the CSR needs three).  Returns the stride (the template's largest valence) if table_host is NULL; otherwise fills
 * table_host (V, stride, 4) for the given stride (>= that valence) and returns MM_OK, or an MMStatus error. */
int mm_build_vertex_corner_table(int32_t V, int32_t F, const int32_t* faces_host, int32_t stride, int32_t* table_host);
const char* mm_status_string(int status);
/* After MM_ERR_LAUNCH on this host thread: "<kernel>: <hipGetErrorString> (hipError n)"; "" if none was recorded. */
const char* mm_last_error_detail(void);
/* Layout guard for bindings that mirror the structs by hand (ctypes, cgo, JNI): sizeof of struct #which as the library was
 * compiled, 0 for an unknown id.  Ids: 0 MMRenderDesc, 1 MMRenderGrads, 2 MMReconDesc, 3 MMMeshRegDesc, 4 MMMeshRegGrads,
 * 5 MMAttLossDesc, 6 MMAttLossGrads, 7 MMTexFlowDesc, 8 MMTexFlowGrads, 9 MMPrepareDesc, 10 MMPrepareGrads, 11 MMDibrDesc,
 * 12 MMDibrGrads, 13 MMTexMapDesc, 14 MMTexMapGrads, 15 MMShDesc, 16 MMShGrads, 17 MMMaskIouDesc. */
size_t mm_struct_size(int which);
/* Bumped whenever a struct or the meaning of a field changes (2: op boundary added, reserved uv-tile fields and profiling slot
 * MM_PROF_BIN removed, options bits defined; 3: MMRenderDesc takes the fixed-stride vertex -> corner table instead of the CSR,
 * MM_OPT_BBOX_MIN_CLOSED_MAX_OPEN; 4: MMRenderDesc.geometry_only / status_flag, MMPrepareDesc.proj_device, MMTexMapGrads.workspace, mm_chamfer_nearest, mm_build_vertex_corner_csr_device; 5: MMRenderDesc.fused_contour; 6: MMRenderDesc.fused_totals, mm_recon_data_totals; still 6: the hint bit MM_OPT_MANY_IN_FLIGHT, which changes no result and no layout).  Bindings must refuse a library whose
 * version differs from what they mirror. */
#define MM_ABI_VERSION 6
int mm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MM_RENDER_H */
