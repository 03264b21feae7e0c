"""AudioConv2dSubsamplingLayer (neurst/layers/modalities/audio_modalities.py:22-109) on the HIP path.

  layer 1  fused conv(3x3,s2,C_in=1)+bias+LayerNorm+ReLU kernel (one HBM pass over the largest activation)
  layer 2  implicit-GEMM conv on MFMA (+bias), then fused LayerNorm+ReLU
  dense    MFMA GEMM [B*T', F'*C] x [F'*C, d]; the PositionEmbeddingWrapper's *sqrt(d) + sinusoid is applied in the
           GEMM epilogue
Variables (TF names/layouts): conv{1,2}/{kernel(HWIO),bias}, ln{1,2}/{gamma,beta}, output_dense/{kernel,bias}.
"""

import torch

from neurst_amd import kernels as K
from neurst_amd.layers.common_layers import Dense, Layer, glorot_uniform


class AudioConv2dSubsamplingLayer(Layer):
    def __init__(self, rt, name, embedding_dim, input_dimension, gen, input_channels=1, channels=256, kernel_size=3,
                 strides=2, layer_norm=True, num_layers=2):
        super().__init__(rt, name)
        if kernel_size != 3 or strides != 2 or num_layers != 2 or input_channels != 1:
            raise NotImplementedError("the HIP front end implements the reference recipe: two 3x3 stride-2 conv "
                                      "layers over single-channel features")
        self._embedding_dim, self._channels, self._layer_norm = embedding_dim, channels, layer_norm
        C, st = channels, rt.store
        self.w1 = st.add(name + "/conv1/kernel", (3, 3, 1, C), glorot_uniform((3, 3, 1, C), gen))
        self.b1 = st.add(name + "/conv1/bias", (C,), torch.zeros(C))
        if layer_norm:
            self.g1 = st.add(name + "/ln1/gamma", (C,), torch.ones(C))
            self.be1 = st.add(name + "/ln1/beta", (C,), torch.zeros(C))
        self.w2 = st.add(name + "/conv2/kernel", (3, 3, C, C), glorot_uniform((3, 3, C, C), gen))
        self.b2 = st.add(name + "/conv2/bias", (C,), torch.zeros(C))
        if layer_norm:
            self.g2 = st.add(name + "/ln2/gamma", (C,), torch.ones(C))
            self.be2 = st.add(name + "/ln2/beta", (C,), torch.zeros(C))
        f2 = ((input_dimension + 1) // 2 + 1) // 2
        self._dense_layer = Dense(rt, name + "/output_dense", f2 * C, embedding_dim, gen)
        # 20 long tiles (K = every encoder row): behind the encoder stack's 240 they would start a second round of the grouped
        # launch (1.05 -> 1.59 ms stand-alone, step 14.08 -> 14.36 ms: profiles/r04_history/c3_group_bench_0.json, c5_ab_step.log); its weight gradient keeps the split-K path on the weight-gradient stream
        self._dense_layer.wgrad_grouped = False   # (as a small group of their own on the weight-gradient stream: 13.0 -> 13.8 ms, c17_ab_side_group.log)
        # (its weight gradient runs at the very end of the backward next to the conv kernels; 512 units measured no better)
        self._dense_layer.wgrad_units = 256

    @property
    def embedding_dim(self):
        return self._embedding_dim

    def forward(self, inputs, timing=None, is_training=True, **kw):
        """inputs [B, T, F, 1] float32 -> [B, T', d] in the compute dtype."""
        assert inputs.dim() == 4 and inputs.shape[-1] == 1
        rt, ln = self.rt, self._layer_norm
        src = inputs.reshape(inputs.shape[0], inputs.shape[1], inputs.shape[2]).float().contiguous()
        B, T, F = src.shape
        a1, mean1, rstd1 = K.conv1_ln_relu_fwd(src, self.w1.data, self.b1.data, self.g1.data if ln else None,
                                               self.be1.data if ln else None, ln, 1e-6, rt.dtype)
        y2 = K.conv2_fwd(a1, self.w2.compute, self.b2.data, relu=not ln)
        _, T2, F2, C = y2.shape
        if ln:
            a2, mean2, rstd2 = K.layernorm_fwd(y2, self.g2.data, self.be2.data, 1e-6, relu=True)
        else:
            a2, mean2, rstd2 = y2, None, None
        d = self._embedding_dim
        epi = {}
        if timing == "sinusoids":
            epi = dict(posenc=rt.posenc(T2, d), posenc_period=T2, emb_scale=float(d) ** 0.5)
        out = self._dense_layer.forward(a2.view(B * T2, F2 * C), **epi)
        if is_training:
            self._saved = (src, a1, mean1, rstd1, y2, a2, mean2, rstd2, timing)
        return out.view(B, T2, d)

    def backward(self, dy, mode="embedding"):
        src, a1, mean1, rstd1, y2, a2, mean2, rstd2, timing = self._saved
        self._saved = None
        st, ln = self.rt.store, self._layer_norm
        B, T2, F2, C = y2.shape
        d = self._embedding_dim
        dz = dy.reshape(B * T2, d)
        if timing == "sinusoids":
            dz = K.scale_dropout_bwd(dz.contiguous(), float(d) ** 0.5, 0.0)
        a2_2d = a2.view(B * T2, F2 * C)
        self._dense_layer.backward_params(a2_2d, dz)
        # (K = 5120 weight gradient next to the dgrad and the LayerNorm backward below; its split-K reduce goes into the same
        # weight-gradient graph -- flushed at the end of the backward pass it started behind the last compute-stream kernel)
        self.rt.flush_wgrads()
        self.rt.sublayer_boundary(force=True)
        if ln:
            da2 = self._dense_layer.backward_input(dz)
            acc = st.acc_flag(self.g2)
            st.acc_flag(self.be2)
            # (the ReLU gate is recomputed from y2 and the saved statistics: a2 -- 295 MB at the benchmark shape -- is not read again)
            if C % 8 == 0 and C <= 1024:
                dy2 = K.layernorm_bwd(da2.view(B, T2, F2, C), y2, self.g2.data, mean2, rstd2, self.g2.grad, self.be2.grad,
                                      accumulate=acc, regate_beta=self.be2.data)
            else:
                dy2 = K.layernorm_bwd(da2.view(B, T2, F2, C), y2, self.g2.data, mean2, rstd2, self.g2.grad, self.be2.grad,
                                      accumulate=acc, y=a2)
        else:  # relu only: gate the dense dgrad with the saved activation
            dy2 = self._dense_layer.backward_input(dz, gate_src=a2_2d, gate_scale=1.0).view(B, T2, F2, C)
        acc2 = st.acc_flag(self.w2)
        assert st.acc_flag(self.b2) == acc2
        da1 = K.conv2_dgrad(dy2, self.w2.compute, a1.shape[1], a1.shape[2])
        # The conv2 weight gradient runs on the COMPUTE stream, behind the data gradient: all three kernels of the front end's
        # backward (conv2 dgrad / wgrad on the 256 x 256 tile core, conv1's backward) hold their CUs alone, so running the weight
        # gradient beside them on the weight-gradient stream only made them take turns -- 2.38 ms for 2.04 ms of stand-alone
        # work; one after the other: 13.10 -> 12.88 ms per step (profiles/r04_history/c36_ab_step.log)
        K.conv2_wgrad(a1, dy2, self.w2.grad, db2=self.b2.grad, accumulate=acc2)
        acc = st.acc_flag(self.w1)
        st.acc_flag(self.b1)
        if ln:
            st.acc_flag(self.g1)
            st.acc_flag(self.be1)
        K.conv1_ln_relu_bwd(src, self.w1.data, self.b1.data, self.g1.data if ln else None,
                            self.be1.data if ln else None, mean1, rstd1, da1, self.w1.grad, self.b1.grad,
                            self.g1.grad if ln else None, self.be1.grad if ln else None, ln, 1e-6, accumulate=acc)
        return None  # the audio features are data, not a differentiable input
