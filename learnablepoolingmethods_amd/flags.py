"""Flag registry with the reference's names and defaults (absl/tf.flags in the reference:
frame_level_models.py:35,2197-2216; video_level_models.py:26-45; train.py:44-112)."""
from __future__ import annotations


class _Flags:
    def __init__(self):
        self.__dict__["_defaults"] = {}

    def define(self, name, default, help=""):  # noqa: A002
        self._defaults[name] = default
        self.__dict__[name] = default

    def reset(self):
        for k, v in self._defaults.items():
            self.__dict__[k] = v

    def __setattr__(self, k, v):
        if k not in self._defaults:
            raise AttributeError(f"unknown flag {k!r}")
        self.__dict__[k] = v


FLAGS = _Flags()
# frame_level_models.py
FLAGS.define("iterations", 30, "Number of frames per batch (frame_level_models.py:35)")
FLAGS.define("sample_random_frames", True, "unused by NetVladV1/V2 (uniform sampling, :2255)")
FLAGS.define("netvlad_add_batch_norm", True, ":2197")
FLAGS.define("netvlad_cluster_size", 256, ":2199")
FLAGS.define("netvlad_hidden_size", 1024, ":2201")
FLAGS.define("netvlad_relu", False, ":2203")
FLAGS.define("gating", True, ":2205")
FLAGS.define("gating_remove_diag", False, ":2207")
FLAGS.define("netvlad_encoder", True, "build extension: False = gated NetVLAD without the cluster encoders (BASELINE cfg-5)")
FLAGS.define("sample_random_frames", True, "frame_level_models.py:40: random frames (True) or a random contiguous sequence")
FLAGS.define("rgb_det_reg", 1e-4, "frame_level_models.py:2213: orthogonality penalty on the rgb cluster centres (WillowModelReg)")
FLAGS.define("audio_det_reg", 1e-4, "frame_level_models.py:2209: orthogonality penalty on the audio cluster centres")
FLAGS.define("netvlad_storage", "f32", "build extension: 'bf16' = the frames, logits / assignment and pooled descriptor of the NetVLAD "
             "streams live in HBM as bf16 with fp32 accumulation (BASELINE configs[4], 'Gated NetVLAD K=512 + MoE-4 ... bf16'); needs "
             "netvlad_encoder off, batch norm on, cluster_size a multiple of 512")
FLAGS.define("netvlad_lazy_descriptor", True, "build extension: NetVladV1's video pooling hands its cluster encoder the un-normalised "
             "residual sums [B, K, D] plus one scale per (clip, cluster); the encoder's block Functions apply the scale where they read "
             "the rows -- the pooled tensor is written once, there is no finalize pass and no transpose in the pooling backward.  "
             "NetVladV2 and the model without cluster encoders (bf16 storage) hand the same form, d-major, to the hidden projection "
             "(ops.projection_parts: the scale applied where the operand is read, no concat)")
FLAGS.define("fused_encoder_blocks", True, "build extension: run the V1 cluster encoder as two block Functions whose backward "
             "folds the gradient sums of shared tensors into GEMM accumulation / the layer-norm kernel (no add passes)")
FLAGS.define("descriptor_slots", True, "build extension: both encoders write their pooled descriptor into one shared buffer "
             "(no concat copy forward, no slice copies backward); needs fused_encoder_blocks")
FLAGS.define("mha_gradient_image", True, "build extension: inside the encoder block Functions the attention kernels write their "
             "result and the q/k/v gradients directly as the split-bf16 operand images of the projection GEMMs (no fp32 copies, no "
             "split passes)")
FLAGS.define("ln_gradient_image", True, "build extension: the FFN block's inner layer-norm backward writes the ReLU-masked "
             "gradient directly as the split-bf16 operand image of the FFN backward GEMMs")
FLAGS.define("input_bn_grad_shortcut", True, "build extension: NetVladV1 training forms input_bn's gamma / beta gradients in closed "
             "form from quantities the pooling backward already has; the [B*S, 1152] input gradient is never computed")
FLAGS.define("ln_pair_forward", True, "build extension: the two layer norms at the end of the V1 encoder share a residual and run "
             "as one three-pass kernel sequence (the first one's output is never stored)")
FLAGS.define("audio_side_stream", True, "build extension: run the audio stream (NetVLAD + encoder, ~100 latency-bound small "
             "launches per step) on a second HIP stream next to the video stream")
FLAGS.define("dense_precision", "bf16x3", "build extension: encoder dense GEMMs as split-bf16 ('bf16x3', ~4e-6) or 'f32'")
FLAGS.define("dense_arithmetic", "fp16x2", "build extension, the encoders' dense GEMMs of NetVladV1 and NetVladV2 (q|k|v, output transform, "
             "both feed-forward layers and every input / weight gradient of them): 'fp16x2' = fp16 (hi, lo) operand planes with per-tensor "
             "power-of-two scales the trainer measures at one step and applies two steps later (ops.OperandScales; the first two steps of a "
             "run stay on split-bf16) -- forward products keep three terms (~1e-6), input gradients two (the weight rounded once to fp16: "
             "1.4e-4 per GEMM), weight gradients one (both operands rounded once: 2e-4, not carried further down the backward) -- against "
             "split-bf16's three terms everywhere; 'bf16x3' = split-bf16 throughout (rounds 1-4)")
# video_level_models.py
FLAGS.define("moe_num_mixtures", 2, "video_level_models.py:27")
FLAGS.define("moe_l2", 1e-8, ":35")
FLAGS.define("moe_low_rank_gating", -1, ":38")
FLAGS.define("moe_prob_gating", False, ":41")
FLAGS.define("moe_prob_gating_input", "prob", ":44")
# train.py
FLAGS.define("batch_size", 1024, "train.py:78")
FLAGS.define("regularization_penalty", 1.0, ":83")
FLAGS.define("base_learning_rate", 0.01, ":86")
FLAGS.define("learning_rate_decay", 0.95, ":88")
FLAGS.define("learning_rate_decay_examples", 4000000, ":91")
FLAGS.define("clip_gradient_norm", 1.0, ":108")
FLAGS.define("hidden1_factored_update", True, "build extension: the GPU trainer consumes hidden1_weights' gradient as the product "
             "descriptors^T . d(activation) it is (lpm_factored_clip_adam): the gradient is never written, the towers all-gather its "
             "two skinny factors instead of all-reducing it.  False: the generic path (gradient written into the arena)")
FLAGS.define("hidden1_compute_copy", True, "build extension, netvlad_storage='bf16' with the factored update: hidden1_weights keeps a bf16 "
             "compute copy beside its fp32 master (SURVEY section 7); the Adam epilogue writes it, the projection's forward and input-gradient "
             "passes read it (2 bytes per weight instead of 4, one bf16 MFMA per product).  False: both stream the fp32 weight (A/B).")
FLAGS.define("hidden1_early_update", True, "build extension, one tower with the factored update: clip + Adam of hidden1_weights run inside backward, "
             "right behind the projection's input gradient (the clip is per variable, utils.py:181-188: it needs only this variable's "
             "gradient factors), where the main queue would otherwise idle while the host enqueues the audio encoder's backward")
FLAGS.define("hidden1_update_stream", "auto", "build extension, with hidden1_early_update: the update pass of hidden1_weights (HBM-bound: 24-26 bytes "
             "per weight) runs on a HIP stream of its own behind the projection's input gradient, UNDER the rest of backward; the step joins it "
             "before it returns.  Measured, alternating on one box: cfg-5 (the gated model: the rest of backward is K3's latency-bound "
             "kernels) 5.495 -> 5.351 ms; cfg-2 (the rest is the encoders' power-limited GEMMs, which the extra HBM traffic slows) 6.64 -> "
             "6.70 ms.  'auto': on for netvlad_storage='bf16' (the configuration whose update pass is half the step), off otherwise; "
             "True / False force it")
FLAGS.define("hidden1_fold_input_gradient", False, "build extension, one tower with hidden1_early_update and the bf16 compute copy: the projection's "
             "INPUT gradient dx = dy W^T is formed inside hidden1_weights' update pass, from the weights that pass streams anyway "
             "(lpm_factored_clip_adam_copy_dx: VERDICT r3-r5), instead of by lpm_proj_dx_w16 from 2 more bytes per weight; the update then runs "
             "on the main stream (the rest of backward waits for dx) whatever hidden1_update_stream says.  Built and measured in round 6, NOT the "
             "default: the pass saves 60-165 us of the two kernels' 2.8 ms box to box and gives up the update stream's overlap with the rest of "
             "backward (~0.14 ms) -- cfg-5 5.431 -> 5.339 ms on one box, 5.275 -> 5.33 ms on another (profiles/r06_update_pass_fold.md).  "
             "LPM_FOLD_DX=0/1 overrides")
FLAGS.define("fold_l2_into_update", True, "build extension, one tower on the GPU: the analytic gradient of the MoE weights' L2 penalties "
             "(coefficient * w) is added inside the clip + Adam passes (lpm_multi_tensor_clip_adam_l2) instead of by an add pass over the "
             "gradient arena in front of them.  False: the add pass (A/B; towers > 1 always add before the all-reduce)")
FLAGS.define("direct_weight_gradients", True, "build extension: single-GPU training writes the encoders' dense-kernel gradients straight "
             "into the gradient arena from their producers (ops._dw_x3) instead of through autograd's .grad + a gather copy")
FLAGS.define("hidden1_factored_max_towers", 4, "build extension: ... up to this many towers.  Both passes of the factored update multiply "
             "over ALL towers' clips (R = 80 N at cfg-2: 0.72 ms at N = 1, 0.94 / 1.2 / 1.73 ms at N = 2 / 4 / 8 measured with "
             "tools/time_factored.py) where the generic route costs 0.88 ms of kernels plus a 554 MB all-reduce under the backward")
FLAGS.define("hidden1_sharded_update", True, "build extension (data parallel, route C of DESIGN.md section 6): beyond hidden1_factored_max_towers "
             "towers hidden1_weights' gradient is reduce-scattered instead of all-reduced, every rank clips (one-float all-reduce of "
             "the shard norms) and Adam-updates its 1/N shard, and the updated shards are all-gathered under the next forward (waited "
             "for where the projection reads the variable).  False: route A (bucket all-reduce + full update on every rank)")
FLAGS.define("hidden1_sharded_min_towers", 0, "build extension: 0 = the sharded route starts right above hidden1_factored_max_towers; "
             "N > 0 = from N towers on, taking precedence over the factored route (tests, A/B on a real node)")
FLAGS.define("library_gemm_selection", True, "build extension: the fp32 library GEMMs the host code leaves to PyTorch (MoE head, context gating, the "
             "small-batch projections) run on the hipBLASLt / rocBLAS solutions recorded per shape in _tunable/gfx950_fp32_gemm.csv (PyTorch's "
             "TunableOp with tuning OFF: a recorded shape takes its recorded solution, any other shape the library's default).  The MoE-4 head of "
             "BASELINE configs[4] is 159 + 87 us forward on the default choices and 47 + 39 us on the recorded ones; the file is ignored (with a "
             "warning from PyTorch) where its ROCm / hipBLASLt / rocBLAS / device validators do not match")
