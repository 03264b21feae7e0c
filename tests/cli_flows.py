"""Command-line flows shared by the GPU tier (tests/test_gpu_model.py) and the host-logic CPU tier
(tests/test_host_path_cpu.py, emulated kernels): one body, so a host-side regression shows up without a GPU box."""
import math

import numpy as np


def cli_tfrecord_flow(tmp_path, REPORT, device=None):
    """The reference's recipe layout end to end (examples/speech_transformer/must-c/st_training_args.yml): a yaml with
    dataset.class AudioTFRecordDataset + task.class SpeechToText (frame-bucketed batches, SpecAugment) drives the trainer
    from TFRecord shards; the loss of the toy model must fall, and model_configs.yml + a checkpoint must be written."""
    import yaml
    import neurst_amd.cli.run_exp as run_exp

    def _run(argv):
        return run_exp._main(argv, device=device)
    from neurst_amd.data import tfrecord
    rng = np.random.RandomState(0)
    data = tmp_path / "train"
    data.mkdir()
    V, fdim = 23, 16
    for i in range(2):
        recs = []
        for _ in range(64):
            frames = int(rng.randint(24, 120))
            tr = rng.randint(0, 4, size=max(2, frames // 12)).astype(np.int64)   # tiny vocabulary: learnable in a few steps
            tr[-1] = V - 1
            recs.append(tfrecord.encode_example({"audio": rng.randn(frames * fdim).astype(np.float32), "translation": tr,
                                                 "uuid": ["u"], "src_lang": ["en"]}))
        tfrecord.write_records(str(data / f"train.tfrecords-{i:05d}-of-00002"), recs)
    cfg = {
        "entry.class": "trainer",
        "entry.params": {"train_steps": 30, "summary_steps": 10, "save_checkpoint_steps": 30,
                         "criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1},
                         "optimizer.class": "adam", "optimizer.params": {"epsilon": 1.e-9, "beta_1": 0.9, "beta_2": 0.98},
                         "lr_schedule.class": "noam", "lr_schedule.params": {"initial_factor": 3.5, "dmodel": 32, "warmup_steps": 10},
                         "validator.class": "CriterionValidator",
                         "validator.params": {"eval_steps": 10, "eval_batch_size": 64, "eval_top_checkpoints_to_keep": 1,
                                              "eval_dataset.class": "AudioTFRecordDataset",
                                              "eval_dataset.params": {"data_path": str(data), "feature_key": "audio",
                                                                      "transcript_key": "translation"}}},
        "dataset.class": "AudioTFRecordDataset",
        "dataset.params": {"data_path": str(data), "shuffle_dataset": True, "feature_key": "audio", "transcript_key": "translation"},
        "task.class": "SpeechToText",
        "task.params": {"audio_feature_dim": fdim, "vocab_size": V, "batch_size": 1200, "max_src_len": 100, "max_trg_len": 12,
                        "min_src_bucket_boundary": 32, "truncate_src": True, "specaug": "SS", "shuffle_buffer": 16},
    }
    cfg_path = tmp_path / "train.yml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    model_dir = tmp_path / "model"
    first = _run(["--config_paths", str(cfg_path), "--hparams_set", "speech_transformer_toy", "--model_dir", str(model_dir),
                           "--dtype", "float32", "--distribution_strategy", "none", "--train_steps", "1"])
    last = _run(["--config_paths", str(cfg_path), "--hparams_set", "speech_transformer_toy", "--model_dir", str(model_dir),
                          "--dtype", "float32", "--distribution_strategy", "none"])
    REPORT["cli_tfrecord.first_loss"], REPORT["cli_tfrecord.last_loss"] = float(first), float(last)
    assert math.isfinite(float(last)) and float(last) < float(first)
    assert (model_dir / "model_configs.yml").exists() and (model_dir / "ckpt-30.index").exists()
    assert (model_dir / "checkpoint").read_text().startswith('model_checkpoint_path: "ckpt-30"')
    # the CriterionValidator ran at steps 10 / 20 / 30 and kept the checkpoint with the best validation NLL
    best = (model_dir / "best" / "checkpoint").read_text()
    assert best.startswith('model_checkpoint_path: "ckpt-') and len(list((model_dir / "best").glob("ckpt-*.index"))) == 1
    # resume: the TensorFlow-format bundle restores weights + Adam state, training continues at step 31
    from neurst_amd.utils import checkpoints as ck
    names = dict(ck.list_variables(str(model_dir / "ckpt-30")))
    assert "SequenceToSequence/input_audio_modality/conv1/kernel" in names   # the reference's default top scope (encoder_decoder_model.py:55-56)
    resumed = _run(["--config_paths", str(cfg_path), "--hparams_set", "speech_transformer_toy", "--model_dir", str(model_dir),
                             "--dtype", "float32", "--distribution_strategy", "none", "--train_steps", "33", "--save_checkpoint_steps", "33"])
    REPORT["cli_tfrecord.resumed_loss"] = float(resumed)
    assert float(resumed) < float(first) and (model_dir / "ckpt-33.index").exists()
    # the "evaluation" entry: NLL / PPL of the restored checkpoint over the same shards, better than chance after training
    ev = _run(["--config_paths", str(cfg_path), "--hparams_set", "speech_transformer_toy", "--model_dir", str(model_dir),
                        "--dtype", "float32", "--distribution_strategy", "none", "--entry", "evaluation", "--batch_size", "40"])
    REPORT["cli_tfrecord.eval_ppl"] = float(ev["PPL"])
    assert set(ev) == {"NLL", "PPL"} and 1.0 < ev["PPL"] < V
    # the "predict" entry (exps/sequence_generator.py): restore the checkpoint, beam-search every utterance, one line each
    out = tmp_path / "hyp.txt"
    hyps = _run(["--config_paths", str(cfg_path), "--hparams_set", "speech_transformer_toy", "--model_dir", str(model_dir),
                          "--dtype", "float32", "--distribution_strategy", "none", "--entry", "predict", "--output_file", str(out),
                          "--batch_size", "50", "--search_method", "beam_search", "--beam_size", "2", "--maximum_decode_length", "12"])
    lines = out.read_text().splitlines()
    assert len(hyps) == len(lines) == 128 and hyps == lines
    ids = [[int(t) for t in line.split()] for line in lines]
    assert all(len(r) <= 12 and all(0 <= t < V - 3 for t in r) for r in ids)          # no EOS / BOS / UNK inside a hypothesis
    assert sum(len(r) for r in ids) > 0
    saved = yaml.safe_load((model_dir / "model_configs.yml").read_text())
    assert saved["task.class"] == "SpeechToText" and saved["task.params"]["audio_feature_dim"] == fdim
