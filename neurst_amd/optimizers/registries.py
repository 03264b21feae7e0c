from neurst_amd.utils.registry import setup_registry

build_optimizer, register_optimizer = setup_registry("optimizer", backend="pt")
build_lr_schedule, register_lr_schedule = setup_registry("lr_schedule", backend="pt")
