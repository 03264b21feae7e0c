"""WordEmbeddingSharedWeights (neurst/layers/modalities/text_modalities.py:21-134) on the HIP path.

mode="embedding": gather (+ *sqrt(d) + sinusoid when wrapped by PositionEmbeddingWrapper) in one kernel;
mode="linear"   : tied logits  x @ W^T + b  on MFMA, reading the [V,d] table as the [N,K] operand (no transpose).
Variables: shared/weights [V,d] + shared/bias [V] when the softmax weights are shared, else emb/weights.
"""
import torch

from neurst_amd import kernels as K
from neurst_amd.layers.common_layers import Layer, _wgrad_split, glorot_uniform


class WordEmbeddingSharedWeights(Layer):
    def __init__(self, rt, name, embedding_dim, vocab_size, gen, share_softmax_weights=False, use_bias=True):
        super().__init__(rt, name)
        self._embedding_dim, self._vocab_size = embedding_dim, vocab_size
        self._share_softmax_weights = share_softmax_weights
        scope = "shared" if share_softmax_weights else "emb"
        init = torch.randn(vocab_size, embedding_dim, generator=gen, dtype=torch.float64) * embedding_dim ** -0.5
        # A vocabulary that is not a multiple of 8 (32 003) makes the table an operand no 16-byte-granular kernel can take: the tied
        # logits, their input gradient and the table's weight gradient then ran on the scalar fall-back kernels (19.6 of the 53 ms
        # of GPU time of a transformer_big step).  The table and the bias therefore keep (Vp8 - V) zero rows behind them in the
        # flat buffers (ParamStore.add tail_pad): the three products run over Vp8 rows / columns, the extra logits are x . 0 + 0,
        # their gradients are kept exactly zero (the criterion zeroes the padding columns of d logits), so the padding never
        # leaves zero and the variables keep their reference shapes [V, d] / [V].
        self._vp8 = (vocab_size + 7) // 8 * 8
        pad_rows = self._vp8 - vocab_size
        self._shared_weights = rt.store.add(f"{name}/{scope}/weights", (vocab_size, embedding_dim), init.float(),
                                            tail_pad=pad_rows * embedding_dim)
        self._bias = None
        if share_softmax_weights and use_bias:
            # created without an initializer in the reference => Keras default glorot_uniform on shape [V]
            self._bias = rt.store.add(f"{name}/{scope}/bias", (vocab_size,), glorot_uniform((vocab_size,), gen), tail_pad=pad_rows)
        self._stack = []

    embedding_dim = property(lambda self: self._embedding_dim)
    vocab_size = property(lambda self: self._vocab_size)

    def forward(self, inputs, mode="embedding", timing=None, is_training=True, time=None, **kw):
        d, V = self._embedding_dim, self._vocab_size
        if mode == "embedding" and time is not None:
            # incremental decoding / streaming encoding (common_layers.py:415-434 with `time`): ids [B'] of ONE position,
            # or [B', n] of the n positions time .. time+n-1; the signal rows start at `time`
            ids = inputs.long()
            ids2 = ids.reshape(-1, 1) if ids.dim() == 1 else ids
            n = ids2.shape[1]
            scale = float(d) ** 0.5 if timing == "sinusoids" else 1.0
            table_len = max(512, 1 << (int(time) + n).bit_length())  # rows do not depend on the table length
            pos = self.rt.posenc(table_len, d)[int(time):int(time) + n].contiguous() if timing == "sinusoids" else None
            out = K.embedding_fwd(self._shared_weights.compute, ids2, pos, n, scale)
            return out.view(-1, d) if ids.dim() == 1 else out
        if mode == "embedding":
            ids = inputs.long()
            L = ids.shape[-1]
            scale = float(d) ** 0.5 if timing == "sinusoids" else 1.0
            pos = self.rt.posenc(L, d) if timing == "sinusoids" else None
            out = K.embedding_fwd(self._shared_weights.compute, ids, pos, L, scale)
            if is_training:
                self._stack.append(("embedding", ids, scale))
            return out
        if mode == "linear":
            x2 = inputs.reshape(-1, d)
            out = None
            esz = x2.element_size()
            if is_training and (V * esz) % 128 != 0:
                # training: the logits are the largest activation of the decoder (9600 x 8008 at the benchmark shape) and are
                # written once and read twice (criterion forward / backward).  A row of V = 8008 bf16 values is 16 016 bytes:
                # rows start 16 bytes further into their 128-byte line each time, every store and load straddles lines.  Rows
                # padded to whole lines (the tensor handed out is the [rows, V] view of the [rows, Vp] buffer)
                vp = ((V * esz + 127) // 128) * (128 // esz)
                out = torch.empty(x2.shape[0], vp, dtype=x2.dtype, device=x2.device)[:, :V]
            vp8 = self._vp8
            if out is not None and vp8 != V and out.stride(0) >= vp8:
                # over the zero-padded table: Vp8 columns into the padded rows, the [rows, V] view handed on (see __init__)
                st = self.rt.store
                wpad, _ = st.padded_views(self._shared_weights, vp8)
                bpad = None if self._bias is None else st.master[self._bias.offset:self._bias.offset + vp8]
                K.gemm(x2, wpad, x2.shape[0], vp8, d, trans_b=True, bias=bpad, out=out.as_strided((x2.shape[0], vp8), out.stride()))
                logits = out
            else:
                logits = K.gemm(x2, self._shared_weights.compute, x2.shape[0], V, d, trans_b=True,
                                bias=None if self._bias is None else self._bias.data, out=out)
            if is_training:
                self._stack.append(("linear", x2))
            return logits.view(*inputs.shape[:-1], V)
        raise ValueError("mode = {} is not valid.".format(mode))

    def backward(self, dy, mode="embedding"):
        st = self.rt.store
        d, V = self._embedding_dim, self._vocab_size
        W = self._shared_weights
        if mode == "linear":
            idx = max(i for i, s in enumerate(self._stack) if s[0] == "linear")
            _, x2 = self._stack.pop(idx)
            dl = dy.reshape(-1, V)
            rows = dl.shape[0]
            acc_w = st.acc_flag(W)
            acc_b = st.acc_flag(self._bias) if self._bias is not None else False

            vp8 = self._vp8
            padded = vp8 != V and dl.stride(0) >= vp8 and getattr(dy, "_nst_zero_padded", 0) >= vp8
            if padded:     # d logits with zero padding columns (the criterion keeps them zero): every product over Vp8
                dlp = dl.as_strided((rows, vp8), dl.stride())
                wpad, gpad = st.padded_views(W, vp8)

                def table_grads():
                    K.gemm(dlp, x2, vp8, d, rows, trans_a=True, out=gpad, accumulate=acc_w, split_k=_wgrad_split(rows, vp8, d, dl.dtype))
                    if self._bias is not None:
                        K.colsum(dl, self._bias.grad, accumulate=acc_b)
                self.rt.run_wgrad(table_grads, dl, x2)
                return K.gemm(dlp, wpad, rows, d, vp8).view(*dy.shape[:-1], d)

            def table_grads():   # parameter gradients only: off the dgrad chain, on the weight-gradient stream
                K.gemm(dl, x2, V, d, rows, trans_a=True, out=W.grad, accumulate=acc_w, split_k=_wgrad_split(rows, V, d, dl.dtype))
                if self._bias is not None:
                    K.colsum(dl, self._bias.grad, accumulate=acc_b)
            self.rt.run_wgrad(table_grads, dl, x2)
            return K.gemm(dl, W.compute, rows, d, V).view(*dy.shape[:-1], d)
        idx = max(i for i, s in enumerate(self._stack) if s[0] == "embedding")
        _, ids, scale = self._stack.pop(idx)
        acc_w = st.acc_flag(W)
        dyc = dy.contiguous()

        def table_grads():   # same stream as the logits' table gradient above: the two accumulate in program order
            if not acc_w:
                W.grad.zero_()
            K.embedding_bwd(dyc, ids, W.grad, scale)
        self.rt.run_wgrad(table_grads, dyc, ids)
        return None
