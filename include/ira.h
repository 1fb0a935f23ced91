/* ira.h -- C ABI of libira_hip.so, the MI355X (gfx950) restoration-inference backend.
 *
 * This is the drop-in boundary for the hot path of blackbean001/Image-Restoration-Agent: the
 * Restormer restoration forward + sliding-window tiler that the agent reaches through
 *   AgenticIR/executor/tool.py:33-78            Tool.__call__ -> _invoke (subprocess "python demo.py ...")
 *   AgentApp/model_service/denoising/Restormer/model_serving.py:158-209   RestormerService.restore_image
 * Plain pointers and sizes only; no torch / C++ types.  Python binds it with ctypes
 * (image-restoration-agent_amd/ira_amd/_capi.py); INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; ira_last_error() gives the text.
 *     Nothing throws or aborts.  There is NO CPU fallback: without a gfx950 device every compute
 *     entry point fails with IRA_ERR_NO_DEVICE.
 *   - "host" pointers are ordinary memory owned by the caller; "dev" pointers are HIP device
 *     allocations on the model's device, also owned by the caller (e.g. a torch.cuda tensor's
 *     data_ptr()).  The library owns weights, workspaces and one HIP stream per model.
 *   - one model handle = one device + one stream, calls on it are serialised by an internal mutex;
 *     handles on different devices are independent (one process per GPU is the intended use).
 *   - all compute entry points are synchronous with respect to the host (they return after the
 *     model stream has drained) unless named *_async.
 *   - image layout: uint8 HWC (interleaved), as cv2/PIL hand it over in the reference
 *     (demo.py:40-41,118).  Network tensors at the ABI: float32 NCHW on the host (what
 *     ``model(input_)`` takes, demo.py:120,131) or float32 NHWC on the device.
 */
#ifndef IRA_H
#define IRA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRA_OK             0
#define IRA_ERR_ARG        1
#define IRA_ERR_NO_DEVICE  2
#define IRA_ERR_HIP        3
#define IRA_ERR_WEIGHTS    4
#define IRA_ERR_OOM        5
#define IRA_ERR_COMM       6   /* a strip-parallel exchange callback returned non-zero, or an RCCL call failed (ira_comm_*) */

typedef struct ira_model ira_model_t;

/* Constructor arguments of the reference Restormer (restormer_arch.py:193-205). hidden[l] is
 * int(dim_l * ffn_expansion_factor) (restormer_arch.py:80), computed by the caller so that the
 * float -> int truncation is done once, in Python, exactly as the reference does it. */
typedef struct ira_restormer_config {
    int32_t inp_channels;            /* 3, or 1 for Gaussian_Gray_Denoising (demo.py:66-67) */
    int32_t out_channels;
    int32_t dim;                     /* 48 */
    int32_t num_blocks[4];           /* {4,6,6,8} */
    int32_t num_refinement_blocks;   /* 4 */
    int32_t heads[4];                /* {1,2,4,8} */
    int32_t hidden[4];               /* int(2.66*dim*2^l): {127,255,510,1021} */
    int32_t ln_with_bias;            /* 0 = BiasFree (denoising tasks, demo.py:61,64,69), 1 = WithBias */
    int32_t conv_bias;               /* 'bias' argument; False in every reference task (demo.py:92) */
    int32_t dual_pixel_task;         /* must be 0 (not dispatched by the agent) */
} ira_restormer_config;

/* Constructor arguments of the reference XRestormer (X-Restormer/xrestormer/archs/xrestormer_arch.py:325-341): the Restormer
 * fields (base.heads = channel_heads) plus the overlapping cross-attention parameters.  Built: window_size 8,
 * overlap_ratio 0.5 (overlap window 12), spatial_dim_head 16, scale 1. */
typedef struct ira_xrestormer_config {
    ira_restormer_config base;
    int32_t spatial_heads[4];        /* {1,2,4,8} in the test presets (options/test/00x_xrestormer_*.yml) */
    int32_t window_size;             /* 8 */
    int32_t overlap_win[4];          /* int(window_size * overlap_ratio) + window_size = 12 */
    int32_t spatial_dim_head;        /* 16 */
    int32_t scale;                   /* 1 */
} ira_xrestormer_config;

/* Constructor arguments of the reference NAFNet (X-Restormer/xrestormer/archs/NAFNet_arch.py:251): SURVEY 8 row a11 /
 * BASELINE config 2.  n_levels = len(enc_blk_nums) == len(dec_blk_nums). */
#define IRA_NAFNET_MAX_LEVELS 6
typedef struct ira_nafnet_config {
    int32_t img_channel;                              /* 3 */
    int32_t width;                                    /* 32: a power of two >= 16 */
    int32_t n_levels;                                 /* 4 */
    int32_t enc_blk_nums[IRA_NAFNET_MAX_LEVELS];      /* {2,2,4,8} */
    int32_t middle_blk_num;                           /* 12 */
    int32_t dec_blk_nums[IRA_NAFNET_MAX_LEVELS];      /* {2,2,2,2} */
} ira_nafnet_config;

/* DnCNN (BASELINE config 1).  The reference contains no DnCNN code: this is the build-defined
 * plumbing model of ira_amd/dncnn.py: conv3x3(in_nc->nc)+ReLU, (nb-2) x [conv3x3(nc->nc)+BatchNorm+ReLU],
 * conv3x3(nc->in_nc), output = input - network(input).  BatchNorm (eval) is folded into the convs by the packer. */
typedef struct ira_dncnn_config {
    int32_t in_nc;     /* 1 (grayscale) or 3 */
    int32_t nc;        /* 64 */
    int32_t nb;        /* 17 conv layers */
} ira_dncnn_config;

/* Constructor arguments of the reference SwinIR (SwinIR/models/network_swinir.py:646-653; task presets inference.py:136-200).
 * ape = False, patch_norm = True, qkv_bias = True as in every reference preset. */
#define IRA_SWINIR_MAX_LAYERS 12
typedef struct ira_swinir_config {
    int32_t upscale;                                  /* 1 (denoise / JPEG), 2, 4 */
    int32_t in_chans;                                 /* 3 or 1 */
    int32_t window_size;                              /* 8, or 7 for the JPEG models */
    float   img_range;                                /* 1.0, or 255.0 for the JPEG models */
    int32_t n_layers;                                 /* len(depths): 6 (M) or 9 (L) */
    int32_t depths[IRA_SWINIR_MAX_LAYERS];            /* 6 each */
    int32_t embed_dim;                                /* 180 (M), 240 (L) */
    int32_t num_heads[IRA_SWINIR_MAX_LAYERS];         /* 6 / 8 each: head_dim 30 */
    int32_t mlp_hidden;                               /* int(embed_dim * mlp_ratio) (:219), computed by the caller */
    int32_t upsampler;                                /* 0 '' (conv_last + input skip), 1 'pixelshuffle', 2 'nearest+conv' */
    int32_t resi_3conv;                               /* 0 '1conv', 1 '3conv' */
} ira_swinir_config;

/* Constructor arguments of the reference MPRNet (AgenticIR/executor/denoising/tools/MPRNet/{Deblurring,Denoising,Deraining}/MPRNet.py:239:
 * 96/48/32, 80/48/32, 40/20/16; kernel_size 3, bias False in all three).  `prelu` is the single parameter of the nn.PReLU() instance every
 * CAB shares (:242). */
typedef struct ira_mprnet_config {
    int32_t in_c, out_c;                              /* 3, 3 */
    int32_t n_feat, scale_unetfeats, scale_orsnetfeats;
    int32_t num_cab;                                  /* 8 */
    int32_t reduction;                                /* 4 */
    float   prelu;
} ira_mprnet_config;

/* Constructor arguments of the reference RRDBNet, the Real-ESRGAN generator (DiffBIR/utils/realesrgan/rrdbnet.py:146; built with
 * 3, 3, scale, 64, 23, 32 by set_realesrgan, realesrganer.py:321-328). */
typedef struct ira_rrdbnet_config {
    int32_t num_in_ch, num_out_ch;
    int32_t scale;                                    /* 4, or 2 / 1 with a pixel-unshuffle of the input (:167-172) */
    int32_t num_feat, num_block, num_grow_ch;         /* 64, 23, 32 */
} ira_rrdbnet_config;

/* ------------------------------------------------------------------ life cycle */

/* Number of visible gfx950 devices. replaces torch.cuda.is_available() (demo.py:98). */
int ira_device_count(int* count);

/* Build a model on `device` from a packed weight blob (ira_amd/pack.py writes it from the
 * reference state_dict: checkpoint['params'], demo.py:101-102). The blob is copied; the caller
 * may free it after the call.  replaces: load_arch['Restormer'](**parameters); model.to(device);
 * model.load_state_dict(...); model.eval()  (demo.py:95-103, model_serving.py:101-124). */
int ira_model_create(int device, const ira_restormer_config* cfg,
                     const void* weight_blob, size_t weight_blob_bytes, ira_model_t** out_model);
/* X-Restormer (blob from ira_amd/xrestormer.py::pack_xrestormer_blob): forward needs H, W multiples of 64; its harness is
 * ira_restore_padcrop_u8 (padding_size 64).  replaces XRestormer(**opt['network_g']) + load_state_dict (xrestormer_arch.py:324-374). */
int ira_xrestormer_create(int device, const ira_xrestormer_config* cfg,
                          const void* weight_blob, size_t weight_blob_bytes, ira_model_t** out_model);

/* Same for NAFNet (blob from ira_amd/nafnet.py::pack_nafnet_blob).  The returned handle is used with the same
 * ira_forward / ira_forward_device / measurement / ira_model_destroy entry points; for NAFNet they accept any H, W
 * (the network zero-pads to a multiple of 2^n_levels and crops: NAFNet_arch.py:303,326,356-361).
 * replaces: NAFNet(**opt); load_state_dict; eval  (NAFNet_arch.py:251-299). */
int ira_nafnet_create(int device, const ira_nafnet_config* cfg,
                      const void* weight_blob, size_t weight_blob_bytes, ira_model_t** out_model);
/* DnCNN: blob from ira_amd/dncnn.py::pack_dncnn_blob; any H, W. */
int ira_dncnn_create(int device, const ira_dncnn_config* cfg,
                     const void* weight_blob, size_t weight_blob_bytes, ira_model_t** out_model);
/* SwinIR: blob from ira_amd/swinir.py::pack_swinir_blob.  ira_forward(_device) then take (B, in_chans, H, W) of any size
 * (reflect-padded to the window inside, network_swinir.py:783-788) and return (B, in_chans, H*upscale, W*upscale). */
int ira_swinir_create(int device, const ira_swinir_config* cfg,
                      const void* weight_blob, size_t weight_blob_bytes, ira_model_t** out_model);
/* MPRNet: blob from ira_amd/mprnet.py::pack_mprnet_blob.  ira_forward(_device) take (B, 3, H, W) with H, W multiples of 8 and return the
 * stage-3 image, the tensor the reference harness keeps (restored = model(input_); restored[0], MPRNet/demo.py:79-80); the harness around
 * it (pad to 8 by reflection, clamp, crop, img_as_ubyte, demo.py:70-87) is ira_restore_u8 with tile = 0.
 * replaces: load_file("MPRNet.py").MPRNet(); utils.load_checkpoint; model.cuda().eval()  (MPRNet/demo.py:57-62). */
int ira_mprnet_create(int device, const ira_mprnet_config* cfg,
                      const void* weight_blob, size_t weight_blob_bytes, ira_model_t** out_model);
/* RRDBNet: blob from ira_amd/rrdbnet.py::pack_rrdbnet_blob.  ira_forward(_device): (B, C, H, W) -> (B, C, H*scale, W*scale), H and W
 * multiples of 2 (scale 2) / 4 (scale 1).  The tiler of RealESRGANer (realesrganer.py:71-183: mod-pad by reflection, tile_process with
 * tile_pad, crop) is ira_restore_padcrop_u8 with padding_size = 2 / 4 / 1.
 * replaces: RRDBNet(...); load_state_dict(loadnet['params_ema' | 'params']); eval  (realesrganer.py:321-328,53-69). */
int ira_rrdbnet_create(int device, const ira_rrdbnet_config* cfg,
                       const void* weight_blob, size_t weight_blob_bytes, ira_model_t** out_model);
int ira_model_destroy(ira_model_t* model);

/* Text of the last error on this handle (or, with model == NULL, of the last failed call that had
 * no handle, e.g. ira_model_create). Valid until the next call on the same handle/thread. */
const char* ira_last_error(const ira_model_t* model);

/* Block until the model stream has drained. */
int ira_sync(ira_model_t* model);

/* ------------------------------------------------------------------ network forward */

/* restored = model(input_)  (demo.py:131,148; restormer_arch.py:245-284).
 * in/out: float32 NCHW host buffers of B*C*H*W elements, C == cfg.inp_channels (== out_channels).
 * H, W must be multiples of 8 (the harness pads, demo.py:122-127). */
int ira_forward(ira_model_t* model, const float* in_nchw_host, int B, int C, int H, int W,
                float* out_nchw_host);

/* Same on device-resident float32 NHWC tensors (B,H,W,C). */
int ira_forward_device(ira_model_t* model, const float* in_nhwc_dev, int B, int H, int W,
                       float* out_nhwc_dev);

/* ------------------------------------------------------------------ harness (demo.py:109-168) */

/* Tiling plan of the overlap-average tiler (demo.py:122-141): padded size (Hp, Wp), effective tile
 * = min(tile, Hp, Wp) (must be a multiple of 8), index lists list(range(0, Hp-tile, stride)) +
 * [Hp-tile].  tile <= 0 means "whole image" (demo.py:129-131): one tile of Hp x Wp.
 * Host-only; any output pointer may be NULL. */
int ira_tile_plan(int h, int w, int tile, int overlap,
                  int* Hp, int* Wp, int* tile_h, int* tile_w, int* n_rows, int* n_cols);

/* Whole harness on host buffers: uint8 HWC in -> uint8 HWC out, h*w*c bytes each.
 * /255 -> reflect-pad-if-needed -> whole / tiled forward -> E/W average -> clamp -> crop ->
 * round-half-even*255.  replaces RestormerService.restore_image minus PNG (de)coding
 * (model_serving.py:158-209) and the per-file body of demo.py:113-161. */
int ira_restore_u8(ira_model_t* model, const uint8_t* in_hwc_host, int h, int w, int c,
                   int tile, int overlap, uint8_t* out_hwc_host);

/* The other tiler of the reference, used by its X-Restormer / HAT style tools (X-Restormer/xrestormer/models/
 * xrestormer_model.py:53-136): reflect-pad bottom/right to a multiple of padding_size (pre_process :53-63), run
 * NON-overlapping tile_size cores, each with a tile_pad halo clipped at the padded frame (:83-95), keep the cores
 * (:119-133), crop (:134-136); then clamp(0,1) -> rint(x*255) (basicsr tensor2img).  tile_size <= 0: untiled (:138-150).
 * Works with every model handle (scale 1).  uint8 HWC host buffers. */
int ira_restore_padcrop_u8(ira_model_t* model, const uint8_t* in_hwc_host, int h, int w, int c,
                           int tile_size, int tile_pad, int padding_size, uint8_t* out_hwc_host);

/* The tiling plan of THIS model's harness: ira_tile_plan's rule for Restormer / NAFNet / DnCNN handles; for a SwinIR handle the
 * always-pad-to-(h // ws + 1) * ws rule of SwinIR/inference.py:76-80,294-310, with `scale` = upscale (tiles_out of
 * ira_restore_tiles_device are then (n, tile_h*scale, tile_w*scale, c) and ira_blend_tiles_device writes (h*scale, w*scale, c)). */
int ira_model_tile_plan(ira_model_t* model, int h, int w, int tile, int overlap,
                        int* Hp, int* Wp, int* tile_h, int* tile_w, int* n_rows, int* n_cols, int* scale);

/* ---- Whole-frame (untiled) Restormer split over the GPUs of a node.  The reference's denoising service runs the
 * frame untiled by default (ModelService/model_services.yaml:74; model_serving.py:170-176), on one GPU.  Here rank r of `world` holds
 * rows [r0, r1) of the (padded) frame, r0 and r1 multiples of 8, plus 8 halo rows of each neighbour; ira_strip_forward_device runs
 * Restormer.forward (restormer_arch.py:245-284) on that band and calls back into the host at the two kinds of exchange step the
 * network has:
 *   exchange(ctx, nbytes):  send send_top[0:nbytes] to rank-1 and send_bot[0:nbytes] to rank+1 (where they exist), receive recv_top
 *                           from rank-1 and recv_bot from rank+1; one row of the feature map before every 3x3 / depthwise 3x3
 *                           (restormer_arch.py:85,108,160,174,184,243): 95 calls per forward.
 *   allgather(ctx, nbytes): gather gather_send[0:nbytes] of every rank into gather_recv, rank-major; the per-block Gram q k^T and the
 *                           squared norms of q and k over this rank's pixels (:121-124): 44 calls per forward.
 * All buffers are device memory owned by the caller (torch tensors); a callback returns 0 once the received bytes are visible to the
 * device.  The host side is ira_amd/dist.py::StripParallelRestorer (torch.distributed over RCCL). */
typedef int (*ira_strip_exchange_fn)(void* ctx, size_t nbytes);
typedef struct ira_strip_comm {
    void* ctx;
    ira_strip_exchange_fn exchange, allgather;
    int32_t rank, world;
    void *send_top, *recv_top, *send_bot, *recv_bot;
    uint64_t edge_capacity;                       /* bytes of each of the four edge buffers: >= Wp * 2 * dim * 4 */
    void *gather_send, *gather_recv;
    uint64_t gather_capacity;                     /* bytes of gather_send (gather_recv holds world times that): >= ira_strip_record_bytes */
} ira_strip_comm;

/* Install (or, with comm == NULL, remove) the exchange callbacks of a Restormer handle. */
int ira_strip_configure(ira_model_t* model, const ira_strip_comm* comm);
/* Largest Gram record (bytes) and largest edge row (bytes, for a padded frame width Wp) the model's forward will stage. */
int ira_strip_buffer_bytes(ira_model_t* model, int Wp, uint64_t* edge_bytes, uint64_t* record_bytes);
/* in / out: float32 NHWC device images of H_local = halo_top + rows + halo_bot rows by W columns; halo_top / halo_bot are 8 where the
 * rank has a neighbour on that side and 0 at the frame border.  Rows [halo_top, H_local - halo_bot) of out are this rank's result. */
int ira_strip_forward_device(ira_model_t* model, const float* in_local_nhwc_dev, int H_local, int W,
                             int halo_top, int halo_bot, float* out_local_nhwc_dev);

/* Sharded form for one-process-per-GPU tile parallelism.  All pointers are device
 * memory.  frame: uint8 (h,w,c), unpadded.  Runs tiles [first_tile, first_tile+n_tiles) of the
 * row-major tile list through the network and writes them, float32 NHWC (n_tiles,tile_h,tile_w,c),
 * to tiles_out.  The caller all-gathers tiles_out across ranks (RCCL) and calls ira_blend_tiles_device. */
int ira_restore_tiles_device(ira_model_t* model, const uint8_t* frame_u8_dev, int h, int w, int c,
                             int tile, int overlap, int first_tile, int n_tiles, float* tiles_out_dev);

/* E/W overlap average in row-major tile order (demo.py:145-153), clamp(0,1), crop to (h,w),
 * rint(x*255) -> uint8 (demo.py:155-161).  tiles_dev holds ALL n_rows*n_cols tiles. */
int ira_blend_tiles_device(ira_model_t* model, const float* tiles_dev, int h, int w, int c,
                           int tile, int overlap, uint8_t* out_u8_dev);

/* ------------------------------------------------------------------ RCCL behind the ABI
 * One process per GPU; the library owns the communicator and issues every collective on the model's own stream (csrc/ira_comm.hip; librccl.so is
 * dlopen-ed by the first of these calls, a single-GPU process never loads it).  The caller only carries the 128-byte id from rank 0 to the
 * other ranks (any side channel: a socket, a file, MPI, torch.distributed's store ...).
 * Reference being replaced: the sequential Python tile loop of demo.py:143-153 (one GPU; AgentApp/agentic_api.py:2 pins it). */
#define IRA_COMM_ID_BYTES 128
int ira_comm_unique_id(void* id_out);                                            /* rank 0: ncclGetUniqueId */
int ira_comm_init(ira_model_t* model, const void* id, int rank, int world);      /* ncclCommInitRank on the model's device */
int ira_comm_destroy(ira_model_t* model);
int ira_comm_rank(ira_model_t* model, int* rank, int* world);

/* One frame, tile-parallel over the communicator's ranks: every rank holds the uint8 (h,w,c) frame in device memory, runs the block
 * [rank*ceil(n/world), ...) of the row-major tile list as one batched forward, ONE ncclAllGather of the float32 tiles (in place, on the model's
 * stream) gives every rank all tiles, and the E/W overlap average runs on every rank in the reference's row-major order (bit-stable): out_dev
 * holds the same uint8 (h*s, w*s, c) frame on every rank.  world == 1 equals ira_restore_tiles_device + ira_blend_tiles_device. */
int ira_restore_frame_distributed(ira_model_t* model, const uint8_t* frame_u8_dev, int h, int w, int c, int tile, int overlap,
                                  uint8_t* out_u8_dev);

/* Band-parallel whole frame (ira_strip_forward_device) with the two exchange steps as ncclSend/ncclRecv and ncclAllGather on the model's
 * stream instead of host callbacks: the library allocates the staging buffers for a padded frame width Wp; rank / world come from the
 * communicator.  Replaces a previous ira_strip_configure. */
int ira_strip_use_comm(ira_model_t* model, int Wp);

/* The whole-frame harness of demo.py:120-131,155-161 (no tiler: model_services.yaml:74) over the communicator's ranks, all on the device and on
 * the model's stream: this rank's band (rows of whole 8-row groups dealt out evenly, + 8 halo rows per neighbour) is cut out of the uint8 frame
 * with the /255 and the pad-if-needed reflect padding, runs through ira_strip_forward_device's schedule with the RCCL exchange steps, its
 * interior rows are clamped / rounded to uint8, and ONE ncclAllGather hands every rank every band: out_u8_dev (h,w,c) is the same on all ranks. */
int ira_strip_restore_u8_device(ira_model_t* model, const uint8_t* frame_u8_dev, int h, int w, int c, uint8_t* out_u8_dev);

/* ------------------------------------------------------------------ measurement */

/* ------------------------------------------------------------------ PNG decode (host only: no device, no handle)
 * replaces cv2.imread / cv2.imdecode in front of the path (AgenticIR/executor/.../Restormer/demo.py:40-44; AgentApp model_serving.py:126-134).
 * 8 bits per sample, colour types 0 / 2 / 3 / 4 / 6, not interlaced; chunk CRCs and the zlib Adler-32 are checked.  ira_png_info reads the
 * header: channels = 1 grey, 2 grey + alpha, 3 RGB (also for palette images, which are expanded), 4 RGBA.  ira_png_decode writes the
 * h x w x channels raster (row-major, RGB order as stored) into out.  Anything else (16-bit, Adam7, 1/2/4-bit) -> IRA_ERR_ARG with the reason
 * in ira_last_error(NULL); the Python wrapper then hands the file to PIL. */
int ira_png_info(const uint8_t* png, size_t len, int* h, int* w, int* channels);
int ira_png_decode(const uint8_t* png, size_t len, uint8_t* out, size_t out_bytes);

/* ------------------------------------------------------------------ brightening tools (no weights, no handle)
 * replaces BrighteningTool._invoke between cv2.imread and cv2.imwrite (AgenticIR/executor/brightening/__init__.py:19-31):
 * cv2.cvtColor(BGR2HSV) -> _update_v(V) -> cv2.cvtColor(HSV2BGR), 8-bit, on RGB-ordered pixels (H, S, V do not depend on the channel order).
 * tool: ConstantShift (:34-41, V + 40 clipped), GammaCorrection (:44-51, gamma 1.5), HistogramEqualization (:54-63, CLAHE clip 2.0, 8 x 8 tiles;
 * frames of at least 16 x 16).  Errors go to ira_last_error(NULL). */
#define IRA_BRIGHTEN_CONSTANT_SHIFT          0
#define IRA_BRIGHTEN_GAMMA_CORRECTION        1
#define IRA_BRIGHTEN_HISTOGRAM_EQUALIZATION  2
int ira_brighten_u8(int device, int tool, const uint8_t* in_rgb_host, int h, int w, uint8_t* out_rgb_host);
/* The same on frames already in HBM (4-byte aligned), enqueued on `hip_stream` (a hipStream_t, NULL = the default stream); in and out may not overlap. */
int ira_brighten_u8_device(int device, void* hip_stream, int tool, const uint8_t* in_rgb_dev, int h, int w, uint8_t* out_rgb_dev);

/* Upper bound on tiles per network launch (workspace is sized for it lazily). 0 = default. */
int ira_set_max_batch(ira_model_t* model, int max_tiles);

/* HIP-event timer on the model stream (the stream the kernels are launched on). */
int ira_timer_start(ira_model_t* model);
int ira_timer_stop(ira_model_t* model, float* elapsed_ms);

/* Per-kernel-family HIP-event timing.  enable(1) makes every launch bracketed by events (slow);
 * ira_profile_json returns {"family": {"launches": n, "ms": t}, ...} accumulated since enable. */
int ira_profile_enable(ira_model_t* model, int on);
const char* ira_profile_json(ira_model_t* model);

/* Bytes of device memory currently held by the model (weights + workspace). */
int ira_memory_bytes(ira_model_t* model, size_t* bytes);

/* ------------------------------------------------------------------ single-kernel entry points
 * Used by the parity tests (tests/test_gpu_ops.py) to check each hand-written kernel against the
 * oracle in isolation.  Device pointers; shapes as documented in DESIGN.md "Kernels".
 * T = number of tiles, P = pixels per tile (H*W). */

/* y[T*P][n_store] (fp16) = LN_mode(x)[T*P][K] @ W[NB*32][K]^T (+ bias).  x is fp32 (in_f32=1) or fp16.
 * ln_mode: 0 none, 1 BiasFree core (x*rstd), 2 WithBias core ((x-mu)*rstd); LN gain/bias are folded
 * into W/bias by the packer.  (restormer_arch.py:37-39,54-57 + the 1x1 convs :82,105) */
int ira_op_pw_kres(ira_model_t* m, const void* x_dev, int in_f32, int ln_mode, int K,
                   const void* W_f16_dev, const float* bias_dev, void* y_f16_dev, int y_stride,
                   int n_store, int NB, int T, int P);

/* y[T*P][N] (fp32) = [x0|x1][T*P][K0+K1] @ W^T (+bias) (+resid).  W is [NB*32][K0+K1] fp16, shared
 * (w_tile_stride = 0) or one matrix per tile (elements between consecutive tiles' matrices).
 * (project_out + residual :131,92,148-149; reduce_chan :223,228; attention apply) */
int ira_op_pw_nres(ira_model_t* m, const void* x0_dev, int K0, const void* x1_dev, int K1, int in_f32,
                   const void* W_f16_dev, long w_tile_stride, const float* bias_dev,
                   const float* resid_dev, float* y_dev, int N, int NB, int T, int P);

/* Depthwise 3x3 (zero pad) on qkv (restormer_arch.py:106,114): in fp16 NHWC (T,H,W,3C), w fp32 [9][3C],
 * bias fp32 [3C] or NULL.  q,k -> fp16 "pixel-minor" (T,H,ceil(W/8),2C,8); v -> fp16 NHWC (T,H,W,C). */
int ira_op_dwconv_qkv(ira_model_t* m, const void* in_dev, const float* w_dev, const float* bias_dev,
                      void* qk_dev, void* v_dev, int T, int H, int W, int C);

/* Depthwise 3x3 + GELU gate (restormer_arch.py:84,90-91): in fp16 NHWC (T,H,W,2*Hp), out fp16 (T,H,W,Hp),
 * out = gelu(dw(in[:Hp])) * dw(in[Hp:]). */
int ira_op_dwconv_gate(ira_model_t* m, const void* in_dev, const float* w_dev, const float* bias_dev,
                       void* out_dev, int T, int H, int W, int Hp);

/* Partial Gram + squared norms (restormer_arch.py:121-124): qk pixel-minor as above ->
 * part fp32 [T][S][heads][c*c + 2c]  (c = C/heads; gram row-major [q ch][k ch], then |q|^2, |k|^2). */
int ira_op_gram(ira_model_t* m, const void* qk_dev, float* part_dev, int T, int nblk, int C, int heads, int S);

/* M[t] = W_o . blockdiag_h(softmax(gram / (|q||k|) * temperature))  -> fp16 [T][NB*32][C]
 * (restormer_arch.py:121-131 re-associated; DESIGN.md "attention as a per-tile 1x1"). */
int ira_op_attn_mix(ira_model_t* m, const float* part_dev, const float* wo_dev, const float* temp_dev,
                    void* M_f16_dev, int T, int C, int heads, int S, int NB);

/* Dense 3x3 conv, zero pad, fp32 NHWC in (T,H,W,K) -> fp32 out.  W fp16 [9][NB*32][K].
 * mode 0: plain (T,H,W,N); 1: + PixelUnshuffle(2) -> (T,H/2,W/2,4N)  (restormer_arch.py:171-179);
 * 2: + PixelShuffle(2) -> (T,2H,2W,N/4) (:181-189); 3: final conv + input skip (:281), skip fp32 (T,H,W,N);
 * 4: plain + ReLU (DnCNN body). */
int ira_op_conv3x3(ira_model_t* m, const float* in_dev, const void* W_f16_dev, const float* bias_dev,
                   const float* skip_dev, float* out_dev, int mode, int T, int H, int W, int K, int N, int NB);

/* Fused GDFN block, x_out = x + ffn(LN(x)) (restormer_arch.py:88-93,149) in one kernel; C in {48, 96}.
 * x, x_out fp32 NHWC (T,H,W,C), must not alias.  Win: packed project_in (LN gain folded, hi/lo planes,
 * fragment-major, hidden padded to Hp), bin fp32 [2*Hp] or NULL; dww fp32 [9][2*Hp], dwb [2*Hp] or NULL;
 * Wout packed project_out, bout fp32 or NULL.  ln_mode 1 = BiasFree, 2 = WithBias. */
int ira_op_ffn_fused(ira_model_t* m, const float* x_dev, float* x_out_dev, int ln_mode, int C,
                     const void* Win_dev, const float* bin_dev, const float* dww_dev, const float* dwb_dev,
                     const void* Wout_dev, const float* bout_dev, int T, int H, int W, int Hp);

/* Fused attention apply + GDFN block, second generation (csrc/ira_ffn2.hip), one kernel; C in {48, 96}:
 *   x1 = x + M_t . v (+ wo_b)   (restormer_arch.py:127-131 re-associated as ira_op_attn_mix does),   x_out = x1 + ffn(LN(x1))  (:88-93,:149).
 * v fp16 NHWC (T,H,W,C) and M (the fp16 hi/lo fragment-major matrices ira_op_attn_mix writes, m_tile_stride halfs per tile) may both
 * be NULL: x1 = x, the plain GDFN block.  dwu fp32 [Hp/16][2][18][8] = the depthwise weights per (16-channel chunk, octet) unit (x1 taps, then gate taps)
 * (ira_amd/pack.py::_dw_unit_pack), dwub fp32 [Hp/16][2][16] or NULL; the other arguments as ira_op_ffn_fused.  x_out must not alias x. */
int ira_op_ffn2(ira_model_t* m, const float* x_dev, float* x_out_dev, const void* v_dev, const void* M_dev, long m_tile_stride,
                const float* wo_b_dev, int ln_mode, int C, const void* Win_dev, const float* bin_dev, const float* dwu_dev,
                const float* dwub_dev, const void* Wout_dev, const float* bout_dev, int T, int H, int W, int Hp);
/* The same with a choice of which 1x1 weights are multiplied as fp16 hi + lo planes (lo_mask bit 0: project_in, bit 1: project_out; 3 =
 * ira_op_ffn2) and which as plain fp16 (their lo plane is not read).  Kernels exist for 3 at C in {48, 96} and for 0 and 1 at C = 96;
 * anything else is IRA_ERR_ARG.  On weights whose lo plane is zero the plain form gives the two-plane form's result bit for bit.  The network
 * schedule runs mask 3 everywhere (the one-plane forms measured 11 % faster but cost parity margin: DESIGN.md 4); the environment variable
 * IRA_FFN_LO_STAGES (eight digits, one mask per Restormer stage, read at create time) selects them for A/B runs. */
int ira_op_ffn2_planes(ira_model_t* m, int lo_mask, const float* x_dev, float* x_out_dev, const void* v_dev, const void* M_dev,
                       long m_tile_stride, const float* wo_b_dev, int ln_mode, int C, const void* Win_dev, const float* bin_dev,
                       const float* dwu_dev, const float* dwub_dev, const void* Wout_dev, const float* bout_dev, int T, int H, int W, int Hp);

/* Which fused GDFN kernel the Restormer schedule uses: 2 (default) = ira_op_ffn2 with the attention apply folded in, 1 = ira_op_ffn_fused
 * after a separate apply kernel.  For A/B measurements; also settable with the environment variable IRA_FFN_GEN at create time. */
int ira_set_ffn_generation(ira_model_t* m, int gen);

/* Weight precision of the LayerNorm-fed and residual 1x1 layers.  1 = plain fp16 weights (the hi plane alone) in the layers that have a
 * kernel for it: the transformer 1x1s of a SwinIR handle; qkv / project_in / project_out / attention apply of the Restormer and X-Restormer
 * levels with C = 192 / 384 (levels 3-4 of the agent's presets).  2 = every weight as fp16 hi + lo planes (two MFMAs per fragment, ~fp32-exact
 * weights).  **The state after create is 1 for Restormer, X-Restormer and SwinIR handles** (the product default, the same for C and Python
 * callers; the other architectures have no hi-plane kernels and ignore the setting).  With 1 the lo planes of those weights are ZEROED in the
 * handle's copy of the blob (after a device-side backup), so that launches too small for the hi-plane kernels give the same numbers (a result
 * never depends on the launch batch); 2 copies the backup back.  Measured: SwinIR-L x4 at the benchmark geometry |dPSNR| 6e-5
 * dB, 149.6 -> 132.8 ms; Restormer 4K frame: emulator 69.4 -> 69.1 dB against the oracle, 235.6 -> 229.1 ms. */
int ira_set_weight_planes(ira_model_t* m, int planes);
int ira_get_weight_planes(ira_model_t* m, int* planes);

/* Fused attention front end: LN + qkv 1x1 + depthwise 3x3 (restormer_arch.py:105-106,114) in one kernel; C in {48, 96}.
 * x fp32 NHWC (T,H,W,C); Wq packed qkv weights (LN gain folded, hi/lo, fragment-major), bq fp32 or NULL; dww fp32
 * [9][3C], dwb [3C] or NULL; outputs as ira_op_dwconv_qkv: q,k pixel-minor (T,H,ceil(W/8),2C,8), v NHWC (T,H,W,C). */
int ira_op_qkv_fused(ira_model_t* m, const float* x_dev, int ln_mode, int C, const void* Wq_dev, const float* bq_dev,
                     const float* dww_dev, const float* dwb_dev, void* qk_dev, void* v_dev, int T, int H, int W);
/* The same op on a named kernel generation (A/B and equivalence tests): 1 = csrc/ira_qkv_fused.hip (one 8-wave workgroup per patch and
 * CU, producer / consumer waves), 3 = csrc/ira_qkv3.hip (4-wave workgroups, two per CU, pixel-stationary transposed 1x1 so that a depthwise
 * lane owns whole pixel-minor elements, weights staged once per workgroup in LDS, MFMAs and depthwise FMAs interleaved).  (Generation 2, a
 * persistent one-workgroup-per-CU kernel that requested the next patch's halo rows under the main loop, was no faster than generation 1 and
 * is not in the library: DESIGN.md 5j.)  ira_op_qkv_fused and the network schedule run generation 3 unless IRA_QKV_GEN=1. */
int ira_op_qkv_fused_gen(ira_model_t* m, int gen, const float* x, int ln_mode, int C, const void* Wq, const float* bq, const float* dww,
                         const float* dwb, void* qk, void* v, int T, int H, int W);

/* Switch the fused kernels (GDFN, attention front end) on (default) or off (per-op kernel path); for A/B measurements.
 * (name kept from the first fused kernel) */
/* Switch the fused GDFN kernel on (default) or off (three-kernel path); for A/B measurements. */
int ira_set_fused_ffn(ira_model_t* m, int on);

/* patch_embed (restormer_arch.py:156-165): fp32 NHWC (T,H,W,ic) -> fp32 (T,H,W,C), W fp32 [C][ic][3][3]. */
int ira_op_embed(ira_model_t* m, const float* in_dev, const float* w_dev, const float* bias_dev,
                 float* out_dev, int T, int H, int W, int ic, int C);

/* torch.cat([a, b], 1) on fp32 NHWC (restormer_arch.py:270). */
int ira_op_concat2(ira_model_t* m, const float* a_dev, int Ca, const float* b_dev, int Cb, float* out_dev, long npx);

/* ira_op_pw_kres with the LayerNorm epsilon as an argument (NAFNet's LayerNorm2d uses 1e-6, NAFNet_arch.py:138;
 * Restormer's 1e-5). */
int ira_op_pw_kres_eps(ira_model_t* m, const void* x_dev, int in_f32, int ln_mode, int K,
                       const void* W_f16_dev, const float* bias_dev, void* y_f16_dev, int y_stride,
                       int n_store, int NB, int T, int P, float eps);

/* ---- X-Restormer kernel: overlapping cross-attention core of OCAB (xrestormer_arch.py:209-232 with RelPosEmb :64-82):
 * qkv fp16 (T,H,W,3*heads*16) = q | k | v (q pre-scaled), rel fp32 [2][32][16] (rel_width rows 0..22, rel_height rows 0..22 of
 * the second tile), out fp16 (T,H,W,heads*16).  8x8 query windows, 12x12 zero-padded key windows. */
int ira_op_ocab_attn(ira_model_t* m, const void* qkv_dev, const float* rel_dev, void* out_dev, int T, int H, int W, int heads);

/* ---- SwinIR kernels (network_swinir.py)
 * (Shifted-)window attention, one call = roll(-shift) + window_partition + WindowAttention core + window_reverse + roll(+shift)
 * (:114-145 inside :239-279): qkv fp16 (T,H,W,3*heads*32) laid out [q|k|v][head][32] (head_dim <= 32 zero-padded; q already
 * scaled), bias fp32 [heads][64][64] = relative_position_bias_table[relative_position_index] as [query][key], out fp16
 * (T,H,W,heads*32).  H, W multiples of ws; ws in {7, 8}; the shift mask of calculate_mask (:216-237) is computed in the kernel. */
int ira_op_swin_attn(ira_model_t* m, const void* qkv_dev, const float* bias_dev, void* out_dev, int T, int H, int W, int heads,
                     int ws, int shift);

/* nn.LayerNorm over the first C of Cp channels, fp32 (npx, Cp) -> fp32; gain/bias fp32 [Cp] (zero beyond C). */
int ira_op_ln_affine(ira_model_t* m, const float* x_dev, const float* g_dev, const float* b_dev, float* y_dev, long npx, int C,
                     int Cp, float eps);

/* ira_op_pw_kres with every option: eps, ln_n = number of real channels when K is zero-padded (0 = K), act 1 = exact-erf GELU. */
int ira_op_pw_kres_ex(ira_model_t* m, const void* x_dev, int in_f32, int ln_mode, int K, const void* W_f16_dev, const float* bias_dev,
                      void* y_f16_dev, int y_stride, int n_store, int NB, int T, int P, float eps, int ln_n, int act);

/* SwinIR MLP half-block in place on the fp32 token stream x (npx, Cp = 256):  x += fc2(GELU(fc1(LayerNorm(x))))
 * (network_swinir.py:274-277, Mlp :14-30; LayerNorm gain/bias folded into fc1 by the packer, ln_n = real channels).  W1 [2][hidp/32][16][64][8],
 * b1 fp32 [hidp], W2 [2][8][hidp/16][64][8], b2 fp32 [256]; the hidden activations stay on chip.  Same result, bit for bit, as
 * ira_op_pw_kres_ex(act 1) into an fp16 (npx, hidp) tensor followed by ira_op_pw_nres on it.  npx >= 32768. */
int ira_op_swin_mlp(ira_model_t* m, float* x_dev, const void* W1_f16_dev, const float* b1_dev, const void* W2_f16_dev, const float* b2_dev,
                    int Cp, int hidp, long npx, float eps, int ln_n);

/* The same with the attention projection + residual in front (network_swinir.py:263-277: the whole second half of a Swin block after the window
 * attention):  x += W_p attn + b_p;  x += fc2(GELU(fc1(LayerNorm(x)))).  attn fp16 (npx, 256) in token order, W_p [2][8][16][64][8], b_p fp32 [256].
 * Same result, bit for bit, as ira_op_pw_nres(attn, ..., resid = x, y = x) followed by ira_op_swin_mlp. */
int ira_op_swin_proj_mlp(ira_model_t* m, float* x_dev, const void* attn_f16_dev, const void* Wp_f16_dev, const float* bp_dev, const void* W1_f16_dev,
                         const float* b1_dev, const void* W2_f16_dev, const float* b2_dev, int Cp, int hidp, long npx, float eps, int ln_n);

/* ---- NAFNet kernels (NAFNet_arch.py:158-326)
 * Pointwise contraction on an MFMA pixel grid of T x Hg x Wg pixels, W packed like every 1x1 here
 * ([2 planes][NBtot][K/16][64][8], per image if w_tile_stride != 0), fp32 result.
 * inmode 0: x fp16 (T,Hg,Wg,K).  1: x fp16 (T,Hg,Wg,2K), operand = x[..,:K]*x[..,K:] (SimpleGate :158-161).
 *        2: x fp32 (T,2Hg,2Wg,K/4), k = (dy*2+dx)*(K/4)+ci: the 2x2 stride-2 "downs" conv (:275-277).  3: x fp32 (T,Hg,Wg,K).
 *        4: x fp16 (T,Hg,Wg,K), operand = x * scale[t] (scale fp32 [T][K]): the channel attention x * sca(x) (:205).
 * outmode 0: y (T,Hg,Wg,N) = acc + bias (+ resid).  1: y (T,2Hg,2Wg,N/4), row n = (2sy+sx)*(N/4)+cc goes to sub-pixel
 *        (sy,sx) channel cc (PixelShuffle(2), :286-291, rows permuted by the packer), + resid (the encoder skip, :320). */
int ira_op_naf_pw(ira_model_t* m, int inmode, int outmode, const void* x_dev, int K, const void* W_f16_dev,
                  long w_tile_stride, const float* bias_dev, const float* resid_dev, float* y_dev, int N, int NBtot,
                  int T, int Hg, int Wg, const float* scale_dev);

/* conv2 (depthwise 3x3, bias) + SimpleGate + partial channel sums for the global average pool (:202-205):
 * in fp16 (T,H,W,2c), w fp32 [9][2c], out fp16 (T,H,W,c), part fp32 [T][blocks_per_image][c] (returned count). */
int ira_op_naf_dw_sg(ira_model_t* m, const void* in_dev, const float* w_dev, const float* bias_dev, void* out_dev,
                     float* part_dev, int T, int H, int W, int c, int* blocks_per_image);

/* Fused NAFBlock tail, in place on x (fp32 (T,P,c), c in {32, 64}):  y = x + W3'(g * s[t]) + b3';
 * x = y + W5'(SimpleGate(W4' LN2d(y) + b4')) + b5'  (:205-219; W3' = diag(beta) conv3, W4' = conv4 diag(norm2 gain),
 * W5' = diag(gamma) conv5, packed hi/lo fragment-major; b* fp32, padded to 32).  g fp16 (T,P,c), s fp32 [T][c]. */
int ira_op_naf_tail(ira_model_t* m, const void* g_dev, const float* s_dev, float* x_dev, int c, const void* W3_dev,
                    const float* b3_dev, const void* W4_dev, const float* b4_dev, const void* W5_dev, const float* b5_dev,
                    int T, int P);

/* s[t] = sca_w . mean_px(g[t]) + sca_b: AdaptiveAvgPool2d(1) + 1x1 conv per image (:175-179); s fp32 [T][c]. */
int ira_op_naf_sca(ira_model_t* m, const float* part_dev, int blocks_per_image, long npx, const float* wsca_dev,
                   const float* bsca_dev, float* s_dev, int T, int c);

#ifdef __cplusplus
}
#endif
#endif /* IRA_H */
