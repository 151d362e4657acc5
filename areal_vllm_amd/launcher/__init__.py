"""Experiment launchers: `python -m areal_vllm_amd.launcher.{local,ray,slurm} <entry.py> --config <yaml> k=v ...`."""
