"""NetVLAD prototypes behind the reference's model-registry API
(reference: frame_level_models.py:2193-2513 NetVladV1 / NetVladV2, :2765-2877 NetVLAD / LightVLAD).

Same names, ``create_model`` signature, variable names and output contract as the reference; the hot
ops (frame sampling + input_bn, soft-assignment GEMM, fused softmax/residual aggregation/normalise,
attention cores) run as hand-written gfx950 kernels through ops.py.  Defect resolutions follow
SURVEY.md App. C (C1 scope_id ignored, C5 tokens = clusters, C8 K // 4, C9 rgb-only input skips audio).
"""
from __future__ import annotations

import contextlib
import math

import torch

from . import FLAGS, layers, model_utils, models, ops, transformer_utils, video_level_models, video_pooling_modules
from . import variables as vs


class NetVLAD():
    """frame_level_models.py:2765-2824.  forward(reshaped_input [(B*max_frames), D]) -> [B, D*K] (d-major)."""

    residual = True

    def __init__(self, feature_size, max_frames, cluster_size, add_batch_norm, is_training, scope_id=None):
        self.feature_size = feature_size
        self.max_frames = max_frames
        self.is_training = is_training
        self.add_batch_norm = add_batch_norm
        self.cluster_size = int(cluster_size)

    def forward(self, reshaped_input, kmajor=False, input_affine=None, storage="f32", lazy=False, out_slot=None):
        """input_affine: (gamma, beta) slices of input_bn when reshaped_input is its (gradient-free) output, see ops.netvlad.
        lazy: hand the k-major descriptor over lazily normalised (ops.netvlad) -- for a consumer that applies the row scale."""
        D, K, dev = self.feature_size, self.cluster_size, reshaped_input.device
        std = 1 / math.sqrt(D)
        cluster_weights = vs.get_variable("cluster_weights", [D, K], vs.random_normal_initializer(std), device=dev)   # :2775
        bn = bias = None
        if self.add_batch_norm:
            bn = layers.bn_variables("cluster_bn", K, dev)                                                          # :2783-2789
        else:
            bias = vs.get_variable("cluster_biases", [K], vs.random_normal_initializer(std), device=dev)            # :2790-2796
        cluster_weights2 = None
        if self.residual:
            cluster_weights2 = vs.get_variable("cluster_weights2", [1, D, K], vs.random_normal_initializer(std),
                                               device=dev)                                                           # :2805-2808
        # matmul -> cluster_bn -> softmax -> a^T x - sum(a) W2 -> l2norm(D) -> flatten -> l2norm  (:2781-2822)
        return ops.netvlad(reshaped_input, cluster_weights, cluster_weights2, self.max_frames, bn=bn, bias=bias,
                           is_training=self.is_training, kmajor=kmajor, input_affine=input_affine, storage=storage, lazy=lazy,
                           out_slot=out_slot)


class LightVLAD(NetVLAD):
    """frame_level_models.py:2827-2877: NetVLAD without the centre-residual term."""
    residual = False


def _project_gate_classify(vlad, vocab_size, cluster_size, hidden1_size, add_batch_norm, relu, gating, remove_diag,
                           is_training, **unused_params):
    """Shared tail of NetVladV1 / NetVladV2: hidden projection, context gating, MoE
    (frame_level_models.py:2309-2377 == :2445-2513)."""
    # vlad = (video, audio | None): the video stream's descriptor LAZILY normalised (ops.vlad_aggregate(lazy=True)) and the concat left to
    # the projection, which reads both blocks where they are (ops.projection_parts)
    parts = vlad if isinstance(vlad, tuple) else None
    dev = parts[0].device if parts else vlad.device
    vlad_dim = sum(p.shape[1] for p in parts if p is not None) if parts else vlad.shape[1]
    hidden1_weights = vs.get_variable("hidden1_weights", [vlad_dim, hidden1_size],
                                      vs.random_normal_initializer(1 / math.sqrt(cluster_size)), device=dev)   # :2315-2317
    if parts and ops.projection_parts_ok(getattr(parts[0], "_lpm_raw", parts[0]), ops.row_scale_of(parts[0]),
                                         getattr(parts[0], "_lpm_scale_ks", 0), parts[1], hidden1_weights):
        activation = ops.projection_parts(parts[0], parts[1], hidden1_weights)                                 # :2309 / :2445 + :2319
    else:
        if parts:
            vlad = ops.materialise(parts[0])
            vlad = torch.cat([vlad, parts[1]], 1) if parts[1] is not None else vlad
        activation = ops.projection(vlad, hidden1_weights) if vlad.is_cuda else vlad.matmul(hidden1_weights)  # :2319
    small = is_training and ops.bn_small_ok(activation)         # clip-level tensors: batch norm + what follows it in one launch each way
    if add_batch_norm and relu and small:
        activation = ops.bn_small(activation, *layers.bn_variables("hidden1_bn", hidden1_size, dev), act=1)    # :2321-2327 + relu6 :2337
    elif add_batch_norm and relu:
        activation = layers.batch_norm(activation, is_training, "hidden1_bn")                                  # :2321-2327
    else:
        hidden1_biases = vs.get_variable("hidden1_biases", [hidden1_size], vs.random_normal_initializer(0.01), device=dev)
        activation = activation + hidden1_biases                                                               # :2329-2334
    if relu and not (add_batch_norm and small):
        activation = torch.clamp(activation, 0.0, 6.0)                                                         # relu6 :2337
    if gating:
        gating_weights = vs.get_variable("gating_weights_2", [hidden1_size, hidden1_size],
                                         vs.random_normal_initializer(1 / math.sqrt(hidden1_size)), device=dev)  # :2343-2346
        gates = activation.matmul(gating_weights)
        if remove_diag:
            gates = gates - torch.diagonal(gating_weights) * activation                                        # :2349-2352
        if not add_batch_norm:
            raise NotImplementedError("context gating without batch norm is broken in the reference (App. C12)")
        if small:
            activation = ops.bn_small(gates, *layers.bn_variables("gating_bn", hidden1_size, dev), act=2, mul=activation)   # :2354-2368
        else:
            gates = layers.batch_norm(gates, is_training, "gating_bn")                                         # :2354-2360
            activation = activation * torch.sigmoid(gates)                                                     # :2367-2368
    vs.summary("activation", activation)
    aggregated_model = getattr(video_level_models, "MoeModel")
    return aggregated_model().create_model(model_input=activation, vocab_size=vocab_size, is_training=is_training,
                                           **unused_params)


class _GammaWatch:
    """The closed-form input_bn gradients divide by gamma (ops._NetVLAD.backward): fine while |gamma| is O(1), inaccurate once an
    element comes within rounding of zero.  This watch keeps min |gamma| under observation WITHOUT stalling the step: a tiny
    reduction + a 4-byte copy into pinned memory every ``EVERY`` calls, read back once its event has completed; below
    ``FLOOR`` the model falls back, for good, to the explicit input-gradient path (same result, ~0.19 ms/step more)."""
    EVERY, FLOOR = 8, 0.2

    def __init__(self):
        self.disabled, self.count, self.pending, self.host = False, 0, None, None

    def ok(self, gamma):
        if self.disabled:
            return False
        if self.count == 0:                                    # first use (fresh or restored weights): decide synchronously, once
            self._decide(float(gamma.detach().abs().min()))
        elif self.pending is not None and self.pending.query():
            self.pending = None
            self._decide(float(self.host))
        if not self.disabled and self.pending is None and self.count % self.EVERY == 0 and self.count > 0:
            if self.host is None:
                self.host = torch.empty((), dtype=torch.float32).pin_memory()
            self.host.copy_(gamma.detach().abs().min(), non_blocking=True)
            self.pending = torch.cuda.Event()
            self.pending.record()
        self.count += 1
        return not self.disabled

    def _decide(self, min_abs_gamma):
        if not (min_abs_gamma >= self.FLOOR):                  # also catches NaN
            import warnings
            self.disabled = True
            warnings.warn(f"input_bn: min |gamma| = {min_abs_gamma:.3g} < {self.FLOOR}: the closed-form gamma / beta gradients are "
                          "switched off, the input gradient is formed explicitly from now on")


def _sample_and_normalise(model_input, num_frames, iterations, add_batch_norm, is_training, storage="f32"):
    """SampleUniformFrames + reshape + input_bn (frame_level_models.py:2248-2271), one fused kernel pair."""
    bn = layers.bn_variables("input_bn", model_input.shape[2], model_input.device) if add_batch_norm else (None,) * 4
    # bf16 storage: the fp32 matrix itself is only filled in when summaries are being collected (nothing else reads it)
    return ops.frame_sample_bn(model_input, num_frames.reshape(-1), iterations, *bn, is_training=is_training, storage=storage,
                               materialize=storage == "f32" or vs.default_store().summaries is not None)


class NetVladV1(models.BaseModel):
    """Paper prototype 1: NetVLAD -> cluster-level transformer encoders -> context gating -> MoE
    (frame_level_models.py:2222-2377)."""

    def create_model(self, model_input, vocab_size, num_frames, iterations=None, add_batch_norm=None,
                     sample_random_frames=None, cluster_size=None, hidden_size=None, is_training=True, encoder=None,
                     **unused_params):
        iterations = iterations or FLAGS.iterations
        add_batch_norm = add_batch_norm or FLAGS.netvlad_add_batch_norm
        cluster_size = cluster_size or FLAGS.netvlad_cluster_size
        hidden1_size = hidden_size or FLAGS.netvlad_hidden_size
        relu, gating, remove_diag = FLAGS.netvlad_relu, FLAGS.gating, FLAGS.gating_remove_diag
        encoder = FLAGS.netvlad_encoder if encoder is None else encoder
        storage = FLAGS.netvlad_storage
        if storage == "bf16" and (encoder or not add_batch_norm or not model_input.is_cuda):
            raise ValueError("netvlad_storage='bf16' is the gated-NetVLAD configuration: netvlad_encoder off, batch norm on, on the GPU")

        max_frames, feature_size = iterations, model_input.shape[2]
        has_audio = feature_size > 1024                                      # App. C9
        shortcut = False
        if (FLAGS.input_bn_grad_shortcut and add_batch_norm and is_training and model_input.is_cuda and torch.is_grad_enabled()
                and ops.netvlad_input_shortcut_ok(max_frames, 1024, cluster_size)
                and (not has_audio or ops.netvlad_input_shortcut_ok(max_frames, 128, cluster_size // 4))):
            # the frames need no gradient of their own, only input_bn's gamma / beta do: the pooling ops take those as inputs and
            # return their gradients in closed form (ops._NetVLAD.backward); the [B*S, 1152] input gradient is never formed
            g_in, b_in, _, _ = layers.bn_variables("input_bn", feature_size, model_input.device)
            watch = getattr(g_in, "_lpm_gamma_watch", None)
            if watch is None:
                watch = g_in._lpm_gamma_watch = _GammaWatch()
            shortcut = watch.ok(g_in)
        if storage == "bf16" and is_training and torch.is_grad_enabled() and not shortcut:
            # bf16 storage writes the frames as operand tiles only and has no input-gradient path: once the closed-form input_bn
            # gradients are off (min |gamma| below the watch's floor, or the flag), this step runs with fp32 storage -- same model,
            # same variables, the explicit gradient path -- instead of reaching the pooling op with unmaterialised frames
            if not getattr(NetVladV1, "_warned_bf16_fallback", False):
                import warnings
                NetVladV1._warned_bf16_fallback = True
                warnings.warn("netvlad_storage='bf16': the closed-form input_bn gradients are unavailable; training steps fall "
                              "back to fp32 storage (explicit input-gradient path)")
            storage = "f32"

        reshaped_input = _sample_and_normalise(model_input, num_frames, iterations, add_batch_norm, is_training, storage)
        if storage == "f32" or vs.default_store().summaries is not None:
            vs.summary("input_bn", reshaped_input)

        video_NetVLAD = NetVLAD(1024, max_frames, cluster_size, add_batch_norm, is_training, "netvlad_rgb_scope")
        audio_NetVLAD = NetVLAD(128, max_frames, cluster_size // 4, add_batch_norm, is_training, "netvlad_audio_scope")
        aff_v = aff_a = None
        if shortcut:
            with torch.no_grad():
                rgb, audio = reshaped_input[:, 0:1024], reshaped_input[:, 1024:]
            if ops.SPLIT_VECTOR and g_in.is_cuda and g_in.dim() == 1 and g_in.shape[0] > 1024:
                (gv, ga), (bv, ba) = ops.split_vector(g_in, 1024), ops.split_vector(b_in, 1024)
                aff_v, aff_a = (gv, bv), (ga, ba)
            else:
                aff_v, aff_a = (g_in[0:1024], b_in[0:1024]), (g_in[1024:], b_in[1024:])
        if aff_v is not None:
            pass
        elif has_audio and reshaped_input.is_cuda:
            rgb, audio = ops.split_columns(reshaped_input, 1024)      # the two slices, sharing one gradient buffer
        else:
            rgb, audio = reshaped_input[:, 0:1024], reshaped_input[:, 1024:]
        # the audio branch is ~100 small, latency-bound launches per step: it runs on a second stream beside the video branch
        # (variables are still created in the reference's order: video_VLAD, audio_VLAD, video_attention, audio_attention)
        use_side = has_audio and reshaped_input.is_cuda and FLAGS.audio_side_stream
        side = ops.side_stream(audio, reshaped_input) if use_side else contextlib.nullcontext()
        # The video descriptor goes to its cluster encoder LAZILY NORMALISED when that encoder will run as block Functions (they apply
        # the per-cluster scale where they read the rows): the pooling then writes the [B, K, D] tensor once and has no finalize pass.
        video_encoder_block = None
        if encoder:
            video_encoder_block = transformer_utils.TransformerEncoder(
                feature_size=1024, hidden_size=1024, num_heads=64, attention_dropout=0.1, ff_filter_size=4 * 1024,
                ff_relu_dropout=0.1, is_train=is_training, scope_id="encode1")
        batch = model_input.shape[0]
        lazy_v = bool(encoder and FLAGS.netvlad_lazy_descriptor and storage == "f32" and reshaped_input.is_cuda
                      and (aff_v is not None or not torch.is_grad_enabled())
                      and video_encoder_block.fused_shape(batch, cluster_size, True) and ops.netvlad_lazy_ok(max_frames, 1024, cluster_size))
        slots = None
        # bf16 storage without the cluster encoders (BASELINE configs[4]): the video descriptor LAZILY normalised -- the bf16 sums as the
        # aggregation kernel wrote them + one scale per (clip, cluster), applied by the projection where it reads them (no finalize pass:
        # that pass read the sums and wrote the fp32 descriptor, 0.4 GB per step at bs 128), the audio descriptor in an fp32 buffer of its own
        lazy5 = bool(storage == "bf16" and not encoder and has_audio and FLAGS.netvlad_lazy_descriptor and FLAGS.descriptor_slots
                     and aff_v is not None and batch <= 128 and hidden1_size % 512 == 0 and cluster_size % 32 == 0)
        if lazy5:
            slots = ops.DescriptorSlots(batch, [(1, 128 * (cluster_size // 4))], reshaped_input)
        elif (storage == "bf16" and not encoder and has_audio and FLAGS.descriptor_slots and aff_v is not None):
            # bf16 storage without the cluster encoders (BASELINE configs[4]): the projection behind the pooling computes in fp32, so the
            # two normalised descriptors leave their finalize passes as fp32 straight into ONE [B, 1024 K + 128 K/4] buffer -- no bf16
            # copy of the descriptor, no concat, no casts of it or of its gradient
            slots = ops.DescriptorSlots(batch, [(1, 1024 * cluster_size), (1, 128 * (cluster_size // 4))], reshaped_input)
        with vs.variable_scope("video_VLAD"):
            vlad_video = video_NetVLAD.forward(rgb, kmajor=encoder, input_affine=aff_v, storage=storage, lazy=lazy_v or lazy5,
                                               out_slot=slots.slots[0] if slots and not lazy5 else None)  # :2273-2274
            if vs.default_store().summaries is not None:
                # [B, K, D] (the App. C5 token view) when the encoders follow, else [B, D*K]
                vs.summary("vlad_video", ops.materialise(vlad_video))
        if has_audio:
            with side, vs.variable_scope("audio_VLAD"):
                vlad_audio = audio_NetVLAD.forward(audio, kmajor=encoder, input_affine=aff_a, storage=storage,
                                                   out_slot=(slots.slots[0 if lazy5 else 1] if slots else None))   # :2276-2277
                vs.summary("vlad_audio", vlad_audio)

        if encoder:
            # tokens = clusters (App. C5): the pooling kernel already wrote the [B, K, D] view
            if has_audio:
                with vs.variable_scope("audio_attention"):
                    audio_encoder_block = transformer_utils.TransformerEncoder(
                        feature_size=128, hidden_size=128, num_heads=16, attention_dropout=0.1, ff_filter_size=4 * 128,
                        ff_relu_dropout=0.1, is_train=is_training, scope_id="encode2")
                if FLAGS.descriptor_slots and video_encoder_block.fused(vlad_video) and audio_encoder_block.fused(vlad_audio):
                    # both encoders write their result straight into one [B, 1024 K + 128 K/4] buffer: the concat below and
                    # the slicing of its gradient cost nothing
                    slots = ops.DescriptorSlots(vlad_video.shape[0], [(cluster_size, 1024), (cluster_size // 4, 128)], vlad_video)
            with vs.variable_scope("video_attention"):
                vlad_video = video_encoder_block.forward(vlad_video, out_slot=slots.slots[0] if slots else None)   # :2282-2292
            if has_audio:
                with side, vs.variable_scope("audio_attention"):
                    vlad_audio = audio_encoder_block.forward(vlad_audio, out_slot=slots.slots[1] if slots else None)  # :2294-2304
            if slots is None:
                vlad_video = vlad_video.reshape(-1, 1024 * cluster_size)
                if has_audio:
                    vlad_audio = vlad_audio.reshape(-1, 128 * (cluster_size // 4))
        if use_side:
            side.join(vlad_audio)

        if lazy5:
            if vs.default_store().summaries is not None:
                vs.summary("vlad", torch.cat([ops.materialise(vlad_video), vlad_audio.reshape(batch, -1)], 1))
            return _project_gate_classify((vlad_video, vlad_audio.reshape(batch, -1)), vocab_size, cluster_size, hidden1_size,
                                          add_batch_norm, relu, gating, remove_diag, is_training, **unused_params)   # :2309 inside the projection
        if slots is not None:
            vlad = slots.join(vlad_video, vlad_audio)                                          # :2309, in place
        else:
            vlad = torch.cat([vlad_video, vlad_audio], 1) if has_audio else vlad_video         # :2309
        vs.summary("vlad", vlad)
        if vlad.dtype != torch.float32:      # bf16 storage: the projection and everything behind it compute in fp32
            vlad = vlad.float()
        return _project_gate_classify(vlad, vocab_size, cluster_size, hidden1_size, add_batch_norm, relu, gating,
                                      remove_diag, is_training, **unused_params)


class WillowModelReg(models.BaseModel):
    """WILLOW model with orthogonal regularisation (frame_level_models.py:2516-2635): random frame
    sampling, input_bn, NetVladOrthoReg on both streams, then the shared projection / context gating / MoE tail."""

    def create_model(self, model_input, vocab_size, num_frames, iterations=None, add_batch_norm=None,
                     sample_random_frames=None, cluster_size=None, hidden_size=None, is_training=True,
                     frame_uniform=None, **unused_params):
        iterations = iterations or FLAGS.iterations
        add_batch_norm = add_batch_norm or FLAGS.netvlad_add_batch_norm
        random_frames = sample_random_frames or FLAGS.sample_random_frames
        cluster_size = cluster_size or FLAGS.netvlad_cluster_size
        hidden1_size = hidden_size or FLAGS.netvlad_hidden_size
        relu, gating, remove_diag = FLAGS.netvlad_relu, FLAGS.gating, FLAGS.gating_remove_diag
        sampler = model_utils.SampleRandomFrames if random_frames else model_utils.SampleRandomSequence
        model_input = sampler(model_input, num_frames.reshape(-1, 1), iterations, uniform=frame_uniform)      # :2539-2544
        max_frames, feature_size = model_input.shape[1], model_input.shape[2]
        reshaped_input = model_input.reshape(-1, feature_size)
        video_NetVLAD = video_pooling_modules.NetVladOrthoReg(1024, max_frames, cluster_size, add_batch_norm, is_training,
                                                              FLAGS.rgb_det_reg, "netvlad_rgb_scope")
        audio_NetVLAD = video_pooling_modules.NetVladOrthoReg(128, max_frames, cluster_size // 4, add_batch_norm, is_training,
                                                              FLAGS.audio_det_reg, "netvlad_audio_scope")
        if add_batch_norm:
            reshaped_input = layers.batch_norm(reshaped_input, is_training, "input_bn")                       # :2558-2564
        has_audio = feature_size > 1024                                                                       # App. C9
        with vs.variable_scope("video_VLAD"):
            vlad = video_NetVLAD.forward(reshaped_input[:, 0:1024])
        if has_audio:
            with vs.variable_scope("audio_VLAD"):
                vlad = torch.cat([vlad, audio_NetVLAD.forward(reshaped_input[:, 1024:])], 1)
        return _project_gate_classify(vlad, vocab_size, cluster_size, hidden1_size, add_batch_norm, relu, gating,
                                      remove_diag, is_training, **unused_params)


class NetVladV2(models.BaseModel):
    """Paper prototype 2: attention-based cluster similarities (frame_level_models.py:2383-2513)."""

    def create_model(self, model_input, vocab_size, num_frames, iterations=None, add_batch_norm=None,
                     sample_random_frames=None, cluster_size=None, hidden_size=None, is_training=True,
                     dropout_masks=None, dropout_rate=None, **unused_params):
        iterations = iterations or FLAGS.iterations
        add_batch_norm = add_batch_norm or FLAGS.netvlad_add_batch_norm
        cluster_size = cluster_size or FLAGS.netvlad_cluster_size
        hidden1_size = hidden_size or FLAGS.netvlad_hidden_size
        relu, gating, remove_diag = FLAGS.netvlad_relu, FLAGS.gating, FLAGS.gating_remove_diag
        dm = dropout_masks or {}

        max_frames, feature_size = iterations, model_input.shape[2]
        has_audio = feature_size > 1024
        split = None
        if (has_audio and add_batch_norm and vs.default_store().summaries is None and ops.frame_sample_bn_split_ok(model_input, 1024)):
            # the two streams' blocks of the sampled, batch-normalised frames as two contiguous matrices straight from the frame-prep
            # kernel (ops.frame_sample_bn_split): same values, same variables; no column slices, copies or gradient concatenation
            bn = layers.bn_variables("input_bn", feature_size, model_input.device)
            split = ops.frame_sample_bn_split(model_input, num_frames.reshape(-1), iterations, *bn, is_training, 1024)
            reshaped_input = None
        else:
            reshaped_input = _sample_and_normalise(model_input, num_frames, iterations, add_batch_norm, is_training)
            vs.summary("input_bn", reshaped_input)

        video_NetVLAD = video_pooling_modules.NetVladAttenCluster(1024, max_frames, cluster_size, add_batch_norm,
                                                                  is_training, "netvlad_rgb_scope")
        audio_NetVLAD = video_pooling_modules.NetVladAttenCluster(128, max_frames, cluster_size // 4, add_batch_norm,
                                                                  is_training, "netvlad_audio_scope")
        if split is not None:
            rgb, audio = split
        elif has_audio and reshaped_input.is_cuda and ops.V2_SPLIT_COLUMNS:
            # one contiguous copy per stream (the encoder and the aggregation both want whole rows); their gradients come back as ONE
            # concatenation instead of two zero-filled [M, 1152] buffers, two slice copies and an add
            # (the copies do not carry the views' shared-gradient slot -- ops._SplitColumns.backward then concatenates the two
            # gradients, which is the point here; the slot mechanism itself serves NetVladV1, whose pooling ops write into it)
            rgb, audio = ops.split_columns(reshaped_input, 1024)
            rgb, audio = rgb.contiguous(), audio.contiguous()
        else:
            rgb, audio = reshaped_input[:, 0:1024], reshaped_input[:, 1024:]
        # (NetVladV1 runs its audio stream on a second HIP stream; here that was measured SLOWER -- 10.95 vs 10.84 ms per step at cfg-3,
        # tools/ab_flags.py: this model's audio stream attends over 300 frames, its launches are long enough to fill the chip by themselves)
        # The video descriptor leaves its pooling LAZILY NORMALISED where the projection can take it that way: the un-normalised sums,
        # written once by the aggregation kernel, + one scale per (clip, cluster) -- no finalize pass, and no tf.concat either (the
        # projection reads the two streams' blocks where they are)
        lazy_v = bool(FLAGS.netvlad_lazy_descriptor and model_input.is_cuda and model_input.shape[0] <= 128 and hidden1_size % 512 == 0
                      and ops.vlad_aggregate_lazy_ok(max_frames, 1024, cluster_size))
        with vs.variable_scope("video_VLAD"):
            vlad_video = video_NetVLAD.forward(rgb, dropout_mask=dm.get("video"), dropout_rate=dropout_rate, lazy=lazy_v)   # :2437-2438
            if vs.default_store().summaries is not None:
                vs.summary("vlad_video", ops.materialise(vlad_video))
        vlad_audio = None
        if has_audio:
            with vs.variable_scope("audio_VLAD"):
                vlad_audio = audio_NetVLAD.forward(audio, dropout_mask=dm.get("audio"), dropout_rate=dropout_rate)  # :2440-2441
                vs.summary("vlad_audio", vlad_audio)
        if lazy_v:
            if vs.default_store().summaries is not None:
                vm = ops.materialise(vlad_video)
                vs.summary("vlad", torch.cat([vm, vlad_audio], 1) if has_audio else vm)
            vlad = (vlad_video, vlad_audio)                                                     # :2445 inside the projection
        else:
            vlad = torch.cat([vlad_video, vlad_audio], 1) if has_audio else vlad_video          # :2445
            vs.summary("vlad", vlad)
        return _project_gate_classify(vlad, vocab_size, cluster_size, hidden1_size, add_batch_norm, relu, gating,
                                      remove_diag, is_training, **unused_params)
