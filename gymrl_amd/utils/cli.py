"""`python -m gymrl_amd.<algorithm>`: the reference's script entry point (ppo_lunarlander.py:429-445 and the same block at
the bottom of every algorithms/*.py): build Config and the trainer, install a SIGINT handler that runs `trainer.test()` and
exits, train, then test.  Engine addition: `--name value` pairs on the command line set Config attributes that exist
(`python -m gymrl_amd.ppo_lunarlander --num_envs 4096 --seed 0`), typed like the attribute's default."""
import signal
import sys


def apply_overrides(config, argv):
    it = iter(argv)
    for tok in it:
        if not tok.startswith("--"):
            raise SystemExit(f"expected --<Config attribute> <value>, got {tok!r}")
        name = tok[2:].replace("-", "_")
        if not hasattr(config, name):
            raise SystemExit(f"{type(config).__name__} has no attribute {name!r}")
        try:
            raw = next(it)
        except StopIteration:
            raise SystemExit(f"--{name} needs a value")
        cur = getattr(config, name)
        if isinstance(cur, bool):
            val = raw.lower() in ("1", "true", "yes", "on")
        elif isinstance(cur, int):
            val = int(float(raw))
        elif isinstance(cur, float):
            val = float(raw)
        elif cur is None:
            val = None if raw.lower() == "none" else (int(raw) if raw.lstrip("-").isdigit() else raw)
        else:
            val = type(cur)(raw)
        setattr(config, name, val)
    return config


def run_script(config_cls, trainer_cls, argv=None, interrupted="\n\nTraining interrupted. Starting test..."):
    config = apply_overrides(config_cls(), sys.argv[1:] if argv is None else argv)
    trainer = trainer_cls(config)

    def signal_handler(signum, frame):
        print(interrupted)
        trainer.test()
        sys.exit(0)

    signal.signal(signal.SIGINT, signal_handler)
    try:
        trainer.train()
    except KeyboardInterrupt:
        print("\nTraining interrupted.")
    trainer.test()
    return trainer
