import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neurst_amd.criterions import build_criterion
from neurst_amd.data.datasets.synthetic_speech import SyntheticSpeechDataset
from neurst_amd.optimizers import build_lr_schedule, build_optimizer
from neurst_amd.tasks import build_task
from neurst_amd.training.train_step import TrainStep
from neurst_amd.utils import compat
from neurst_amd.utils.hparams_sets import get_hyper_parameters
dev = "cuda:0"
hp = get_hyper_parameters("speech_transformer_s")
B, T, F, V = 128, 900, 80, 8008
L = T // 12
task = build_task({"task.class": "speech2text", "task.params": {"audio_feature_dim": F, "vocab_size": V}})
model = task.build_model(hp, device=dev, dtype="bfloat16", seed=1234, init_seed=42)
crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
opt = build_optimizer({"optimizer.class": hp["optimizer.class"], "optimizer.params": hp["optimizer.params"]})
opt.bind(model.store)
opt.learning_rate = build_lr_schedule({"lr_schedule.class": hp["lr_schedule.class"], "lr_schedule.params": hp["lr_schedule.params"]})
step = TrainStep(model, crit, opt, None)
ds = SyntheticSpeechDataset({"batch_per_gpu": B, "frames": T, "feature_dim": F, "trg_len": L, "vocab_size": V, "ragged": False, "seed": 1234})
it = ds.build_iterator(map_func=lambda b: task.example_to_input(b, compat.ModeKeys.TRAIN), shard_id=0, total_shards=1, device=dev)
batch = next(it)
for _ in range(3): step(batch)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N): step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/N:.2f} ms/step ; total {1e3*(t2-t0)/N:.2f} ms/step ; GPU tail after last enqueue {1e3*(t2-t1):.2f} ms")
