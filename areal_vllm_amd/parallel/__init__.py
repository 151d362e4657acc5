"""Parallelism: process grid, ZeRO sharded optimizer, tensor/sequence/pipeline/expert parallel helpers."""
