#!/usr/bin/env python3
"""Extracts the golden vectors that pin the CPU oracle from the reference's own tests.

Runs ONLY in the build container (it reads /root/reference, which does not exist
on the GPU box); the resulting ``*.npz`` fixtures are committed next to this
script and are what ``tests/`` loads.  No reference source is copied: the script
parses the reference test files with ``ast`` and pulls out the *literal arrays*
(inputs, pinned kernels, expected outputs) -- the numbers, not the code.

  python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Sources (all under /root/reference/tests):
  neurst/layers/attentions/multi_head_attention_test.py:7-111
  neurst/layers/encoders/transformer_encoder_test.py:21-122
  neurst/layers/decoders/transformer_decoder_test.py:20-158
  neurst/layers/common_layers_test.py:96-152   (sinusoid position embedding)
  neurst/models/transformer_test.py:23-666     (2+2 layer enc-dec logits)
Additionally (``gen_neurst_pt_frontend``) the reference's own PyTorch mirror
``neurst_pt`` is imported under a small shim and executed to produce
input/output vectors for the conv front-end and the 1+1-layer
``speech_transformer_toy`` forward (neurst_pt/layers/modalities/audio_modalities.py:22-100,
neurst_pt/models/speech_transformer.py), which have no literal golden vector.
"""
import ast
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------
# AST helpers
# ----------------------------------------------------------------------------
def _call_name(node):
    if not isinstance(node, ast.Call):
        return None
    f = node.func
    parts = []
    while isinstance(f, ast.Attribute):
        parts.append(f.attr)
        f = f.value
    if isinstance(f, ast.Name):
        parts.append(f.id)
    return ".".join(reversed(parts))


def _dtype_of(call):
    for kw in call.keywords:
        if kw.arg == "dtype":
            return ast.unparse(kw.value)
    if len(call.args) > 1:
        return ast.unparse(call.args[1])
    return None


def literal(node):
    """np.ndarray for tf.convert_to_tensor(<lit>) / numpy.array(<lit>), else None."""
    name = _call_name(node)
    if name in ("tf.convert_to_tensor", "numpy.array", "np.array"):
        inner = node.args[0]
        got = literal(inner)
        if got is None:
            try:
                got = np.array(ast.literal_eval(inner))
            except Exception:
                return None
        dt = _dtype_of(node) or ""
        if "int" in dt:
            return got.astype(np.int64)
        if "float" in dt:
            return got.astype(np.float32)
        return got.astype(np.float32) if got.dtype.kind == "f" else got
    return None


def find_function(tree, name):
    for n in tree.body:
        if isinstance(n, ast.FunctionDef) and n.name == name:
            return n
    raise KeyError(name)


def assigned_literals(fn):
    out = {}
    for n in ast.walk(fn):
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name):
            lit = literal(n.value)
            if lit is not None and n.targets[0].id not in out:
                out[n.targets[0].id] = lit
            elif isinstance(n.value, ast.Dict):
                for k, v in zip(n.value.keys, n.value.values):
                    lv = literal(v)
                    if lv is not None and isinstance(k, ast.Constant):
                        out.setdefault(f"{n.targets[0].id}.{k.value}", lv)
    return out


def _first_literal_in(nodes):
    for b in nodes:
        for n in ast.walk(b):
            lit = literal(n)
            if lit is not None:
                return lit
    return None


def wname_literals(fn, stop_lineno=None):
    """{"substring of w.name": array} for `if "<str>" in w.name: assign(w, <lit>)` chains."""
    out = {}
    for n in ast.walk(fn):
        if isinstance(n, ast.If) and isinstance(n.test, ast.Compare) and len(n.test.ops) == 1 \
                and isinstance(n.test.ops[0], ast.In) and isinstance(n.test.left, ast.Constant) \
                and isinstance(n.test.left.value, str):
            if stop_lineno is not None and n.lineno > stop_lineno:
                continue
            lit = _first_literal_in(n.body)
            if lit is not None and n.test.left.value not in out:
                out[n.test.left.value] = lit
    return out


def wshape_literals(fn):
    """{shape tuple: array} for `if w.shape == (..): predefined.append(<lit>)` chains."""
    out = {}
    for n in ast.walk(fn):
        if isinstance(n, ast.If) and isinstance(n.test, ast.Compare) and len(n.test.ops) == 1 \
                and isinstance(n.test.ops[0], ast.Eq) and isinstance(n.test.comparators[0], ast.Tuple):
            shape = tuple(ast.literal_eval(n.test.comparators[0]))
            lit = _first_literal_in(n.body)
            if lit is not None and shape not in out:
                out[shape] = lit
    return out


def assert_literals(fn):
    """Literal arrays appearing inside assert statements / assert_equal_numpy calls, in source order."""
    found = []
    for n in ast.walk(fn):
        target = None
        if isinstance(n, ast.Assert):
            target = n.test
        elif isinstance(n, ast.Expr) and _call_name(n.value) == "assert_equal_numpy":
            target = n.value
        if target is None:
            continue
        for m in ast.walk(target):
            lit = literal(m)
            if lit is not None and _call_name(m) != "tf.convert_to_tensor":
                found.append((m.lineno, lit))
                break
    found.sort(key=lambda t: t[0])
    return [f[1] for f in found]


def parse(rel):
    with open(os.path.join(REF, rel)) as fp:
        return ast.parse(fp.read())


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **arrays)
    print(f"wrote {path}: " + ", ".join(f"{k}{list(v.shape)}" for k, v in arrays.items()))


# ----------------------------------------------------------------------------
# literal golden vectors
# ----------------------------------------------------------------------------
def gen_attention():
    tree = parse("tests/neurst/layers/attentions/multi_head_attention_test.py")
    fn = find_function(tree, "test_multihead_attention")
    a, s, e = assigned_literals(fn), wshape_literals(fn), assert_literals(fn)
    save("mha_cross", query=a["query"], memory=a["memory"],
         **{"w:output_transform/kernel": s[(4, 3)], "w:output_transform/bias": s[(3,)],
            "w:q_transform/kernel": s[(1, 4)], "w:q_transform/bias": s[(4,)],
            "w:kv_transform/kernel": s[(1, 8)], "w:kv_transform/bias": s[(8,)]},
         expected=e[0], num_heads=np.array(2))
    fn = find_function(tree, "test_multiheadself_attention")
    a, s, e = assigned_literals(fn), wshape_literals(fn), assert_literals(fn)
    save("mha_self", query=a["query"], bias=a["bias"],
         **{"w:output_transform/kernel": s[(4, 3)], "w:output_transform/bias": s[(3,)],
            "w:qkv_transform/kernel": s[(2, 12)], "w:qkv_transform/bias": s[(12,)]},
         expected=e[0], num_heads=np.array(2))


def gen_encoder():
    tree = parse("tests/neurst/layers/encoders/transformer_encoder_test.py")
    fn = find_function(tree, "test_transformer_encoder")
    a, w, e = assigned_literals(fn), wname_literals(fn), assert_literals(fn)
    save("transformer_encoder", inputs=a["inputs"], input_padding=a["input_padding"],
         **{"w:TransformerEncoder/" + k: v for k, v in w.items()},
         expected=e[0], num_heads=np.array(2), num_layers=np.array(1))


def gen_decoder():
    tree = parse("tests/neurst/layers/decoders/transformer_decoder_test.py")
    fn = find_function(tree, "test_transformer_decoder")
    a, w, e = assigned_literals(fn), wname_literals(fn), assert_literals(fn)
    save("transformer_decoder", encoder_outputs=a["encoder_outputs"],
         encoder_inputs_padding=a["encoder_inputs_padding"], decoder_inputs=a["decoder_inputs"],
         **{"w:TransformerDecoder/" + k: v for k, v in w.items()},
         expected=e[0], num_heads=np.array(2), num_layers=np.array(1))


def gen_position_embedding():
    tree = parse("tests/neurst/layers/common_layers_test.py")
    fn = find_function(tree, "test_position_embedding")
    a, e = assigned_literals(fn), assert_literals(fn)
    table = None
    for n in ast.walk(fn):  # embedding_layer.set_weights([numpy.array(...)])
        if _call_name(n) == "embedding_layer.set_weights":
            table = _first_literal_in([n])
            break
    save("position_embedding", inputs1d=a["inputs1d"], inputs2d=a["inputs2d"], table=table,
         expected_2d=e[0], expected_1d_time3=e[1])


def gen_transformer():
    tree = parse("tests/neurst/models/transformer_test.py")
    fn = find_function(tree, "test_seq2seq")
    e = assert_literals(fn)
    first_assert_line = min(n.lineno for n in ast.walk(fn) if isinstance(n, ast.Assert))
    a, w = assigned_literals(fn), wname_literals(fn, stop_lineno=first_assert_line)
    save("transformer_toy_logits",
         src=a["parsed_inputs.src"], src_padding=a["parsed_inputs.src_padding"],
         trg_input=a["parsed_inputs.trg_input"], trg=a["parsed_inputs.trg"],
         trg_padding=a["parsed_inputs.trg_padding"],
         **{"w:" + k: v for k, v in w.items()},
         expected=e[0], num_heads=np.array(2), num_layers=np.array(2))


# ----------------------------------------------------------------------------
# vectors generated by executing the reference's own neurst_pt code
# ----------------------------------------------------------------------------
def _install_shim():
    """Minimal import shim so the UNMODIFIED neurst_pt layer files import without
    tensorflow/absl: tf.nest helpers, absl.logging, and stub
    neurst.utils.{registry,flags_core,configurable,compat}."""
    import logging as pylog

    def flatten(x):
        if isinstance(x, (list, tuple)):
            return [z for y in x for z in flatten(y)]
        if isinstance(x, dict):
            return [z for k in sorted(x) for z in flatten(x[k])]
        return [x]

    def is_nested(x):
        return isinstance(x, (list, tuple, dict))

    def map_structure(fn, *xs):
        x0 = xs[0]
        if isinstance(x0, (list, tuple)):
            return type(x0)(map_structure(fn, *ys) for ys in zip(*xs))
        if isinstance(x0, dict):
            return {k: map_structure(fn, *[x[k] for x in xs]) for k in x0}
        return fn(*xs)

    def pack_sequence_as(structure, flat):
        flat = list(flat)

        def rec(s):
            if isinstance(s, (list, tuple)):
                return type(s)(rec(y) for y in s)
            if isinstance(s, dict):
                return {k: rec(s[k]) for k in sorted(s)}
            return flat.pop(0)
        return rec(structure)

    tf = types.ModuleType("tensorflow")
    tf.nest = types.SimpleNamespace(flatten=flatten, is_nested=is_nested, map_structure=map_structure,
                                    pack_sequence_as=pack_sequence_as)
    sys.modules["tensorflow"] = tf
    absl = types.ModuleType("absl")
    absl.logging = pylog.getLogger("absl")
    sys.modules["absl"] = absl
    sys.modules["absl.logging"] = absl.logging

    neurst = types.ModuleType("neurst")
    neurst.__path__ = []
    utils = types.ModuleType("neurst.utils")
    utils.__path__ = []
    registry = types.ModuleType("neurst.utils.registry")

    def setup_registry(name, base_class=None, create_fn=None, verbose_creation=False, backend="tf"):
        def build(*a, **k):
            raise NotImplementedError

        def register(x):
            if callable(x) and not isinstance(x, (str, list)):
                return x
            return lambda c: c
        return build, register
    registry.setup_registry = setup_registry
    registry.REGISTRIES = {}
    flags_core = types.ModuleType("neurst.utils.flags_core")

    class Flag:
        TYPE = types.SimpleNamespace(INTEGER=int, BOOLEAN=bool, FLOAT=float, STRING=str)

        def __init__(self, name, dtype=None, default=None, help="", **kw):
            self.name, self.default = name, default

    class ModuleFlag(Flag):
        def __init__(self, name, module_name=None, default=None, help=""):
            self.name, self.default = name, default
    flags_core.Flag, flags_core.ModuleFlag = Flag, ModuleFlag
    configurable = types.ModuleType("neurst.utils.configurable")
    configurable.extract_constructor_params = lambda loc, verbose=False: {}
    compat = types.ModuleType("neurst.utils.compat")
    compat.FLOAT_MIN = -1.e9
    for m, n in ((neurst, "neurst"), (utils, "neurst.utils"), (registry, "neurst.utils.registry"),
                 (flags_core, "neurst.utils.flags_core"), (configurable, "neurst.utils.configurable"),
                 (compat, "neurst.utils.compat")):
        sys.modules[n] = m
    # neurst_pt as a namespace package whose __init__ files are NOT executed
    for pkg in ("neurst_pt", "neurst_pt.layers", "neurst_pt.layers.modalities", "neurst_pt.layers.attentions",
                "neurst_pt.layers.encoders", "neurst_pt.layers.decoders", "neurst_pt.utils"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, *pkg.split("."))]
        sys.modules[pkg] = m


def _load(modname):
    import importlib.util
    path = os.path.join(REF, *modname.split(".")) + ".py"
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def gen_neurst_pt_frontend():
    import torch
    _install_shim()
    am = _load("neurst_pt.layers.modalities.audio_modalities")
    torch.manual_seed(7)
    rng = np.random.RandomState(7)
    for tag, (B, T, Fdim, C, d, ln) in {"frontend_ln": (2, 11, 80, 5, 8, True),
                                         "frontend_noln": (1, 14, 12, 6, 4, False),
                                         "frontend_ragged": (3, 23, 16, 8, 8, True)}.items():
        layer = am.AudioConvSubsamplingLayer(embedding_dim=d, input_dimension=Fdim, input_channels=1,
                                             channels=C, layer_norm=ln)
        if ln:  # make LN affine non-trivial
            for l in (layer._norm_layer1, layer._norm_layer2):
                l.weight.data = torch.tensor(rng.uniform(0.5, 1.5, C), dtype=torch.float32)
                l.bias.data = torch.tensor(rng.uniform(-0.3, 0.3, C), dtype=torch.float32)
        src = rng.randn(B, T, Fdim, 1).astype(np.float32)
        with torch.no_grad():
            out = layer(torch.tensor(src)).numpy()
        # inverse of the TF->PT map of tests/neurst_pt/modalities/audio_modalities_test.py:32-37
        w = {"w:input_audio_modality/conv1/kernel": layer._conv_layer1.weight.data.numpy().transpose(2, 3, 1, 0),
             "w:input_audio_modality/conv1/bias": layer._conv_layer1.bias.data.numpy(),
             "w:input_audio_modality/conv2/kernel": layer._conv_layer2.weight.data.numpy().transpose(2, 3, 1, 0),
             "w:input_audio_modality/conv2/bias": layer._conv_layer2.bias.data.numpy(),
             "w:input_audio_modality/output_dense/kernel": layer._dense_layer.weight.data.numpy().T,
             "w:input_audio_modality/output_dense/bias": layer._dense_layer.bias.data.numpy()}
        if ln:
            w.update({"w:input_audio_modality/ln1/gamma": layer._norm_layer1.weight.data.numpy(),
                      "w:input_audio_modality/ln1/beta": layer._norm_layer1.bias.data.numpy(),
                      "w:input_audio_modality/ln2/gamma": layer._norm_layer2.weight.data.numpy(),
                      "w:input_audio_modality/ln2/beta": layer._norm_layer2.bias.data.numpy()})
        save("neurst_pt_" + tag, src=src, expected=out, layer_norm=np.array(int(ln)),
             **{k: np.ascontiguousarray(v) for k, v in w.items()})


def main():
    gen_attention()
    gen_encoder()
    gen_decoder()
    gen_position_embedding()
    gen_transformer()
    gen_neurst_pt_frontend()


if __name__ == "__main__":
    main()
