"""``aggfly`` command (`aggfly/cli/main.py:18-255`): info | regions | validate | weights | run."""
from __future__ import annotations

import sys

import click

from . import config as cfg
from . import pipeline


@click.group()
def cli():
    """Aggregate gridded climate data onto regions on AMD MI355X GPUs."""


@cli.command()
@click.argument("path")
@click.option("--var", default=None, help="variable to describe")
def info(path, var):
    """Describe a dataset: dims, dtype, time span, calendar, longitude convention."""
    import json
    import os
    from aggfly_amd.io import _looks_like_zarr
    if _looks_like_zarr(path):
        arrays = sorted(d for d in os.listdir(path) if os.path.exists(os.path.join(path, d, ".zarray")))
        click.echo(f"zarr store: {path}\narrays: {', '.join(arrays)}")
        for a in arrays if var is None else [var]:
            meta = json.load(open(os.path.join(path, a, ".zarray")))
            attrs = json.load(open(os.path.join(path, a, ".zattrs"))) if os.path.exists(os.path.join(path, a, ".zattrs")) else {}
            click.echo(f"  {a}: shape={meta['shape']} chunks={meta['chunks']} dtype={meta['dtype']} "
                       f"dims={attrs.get('_ARRAY_DIMENSIONS')} units={attrs.get('units')} calendar={attrs.get('calendar')}")
    if var is not None:
        import aggfly_amd as af
        import numpy as np
        ds = af.dataset_from_path(path, var)
        lon = ds.longitude
        click.echo(f"{var}: {ds.da.sizes} dtype={ds.da.dtype}")
        click.echo(f"time: {ds.time[0]} .. {ds.time[-1]} ({len(ds.time)} steps, calendar={getattr(ds.time, 'calendar', 'standard')})")
        click.echo(f"longitude: {lon.min():.3f} .. {lon.max():.3f} -> lon_is_360: {bool(np.nanmax(lon) > 180)}")
        click.echo(f"latitude: {ds.latitude.min():.3f} .. {ds.latitude.max():.3f}")


@cli.command()
@click.argument("path")
@click.option("--regionid", default=None)
def regions(path, regionid):
    """List the columns / ids of a region table."""
    from aggfly_amd.weights import _read_table
    t = _read_table(path)
    click.echo(f"{len(t)} regions; columns: {list(t.columns)}")
    if regionid:
        click.echo(f"{regionid}: {t[regionid].head(10).tolist()} ...")


@cli.command()
@click.argument("config_path")
def validate(config_path):
    """Validate a YAML config without touching the data."""
    try:
        c = cfg.load_config(config_path)
    except cfg.ConfigError as e:
        click.echo(f"Invalid config:\n{e}", err=True)
        sys.exit(1)
    click.echo(f"OK: {len(c.variables)} variable(s), {len(c.resolved_paths())} dataset path(s), engine={c.engine}")


@cli.command()
@click.argument("config_path")
def weights(config_path):
    """Locate and summarise the precomputed weights a config will use."""
    try:
        c = cfg.load_config(config_path)
        w, _, _ = pipeline.compute_weights(c, click.echo)
    except (cfg.ConfigError, FileNotFoundError) as e:
        click.echo(str(e), err=True)
        sys.exit(1)
    t = w.weights
    click.echo(f"{len(t)} (cell, region) pairs; {t['index_right'].nunique()} regions; zero_weight={w.zero_weight}")


@cli.command()
@click.argument("config_path")
@click.option("--engine", type=click.Choice(sorted(cfg.ALLOWED_ENGINE)), default=None, help="override aggregate.engine")
@click.option("--years", default=None, help="override years ('start:end' or a single year)")
@click.option("--output", "output_path", default=None, help="override output.path")
@click.option("--quiet", is_flag=True)
def run(config_path, engine, years, output_path, quiet):
    """Run the whole pipeline and write the panel."""
    try:
        c = cfg.load_config(config_path)
    except cfg.ConfigError as e:
        click.echo(f"Invalid config:\n{e}", err=True)
        sys.exit(1)
    if engine:
        c.engine = engine
    if years:
        errs = []
        c.years = cfg._parse_years(years, errs)
        if errs:
            click.echo("\n".join(errs), err=True)
            sys.exit(1)
    if output_path:
        c.output_path = output_path
    _maybe_init_distributed()
    log = (lambda m: None) if quiet else click.echo
    df = pipeline.run_pipeline(c, log)
    from aggfly_amd.distributed import world
    if world()[0] == 0:
        pipeline.write_output(df, c.output_path, c.output_format)
        log(f"Wrote {len(df)} rows x {len(df.columns)} columns to {c.output_path}")


def _maybe_init_distributed():
    import os
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            ndev = torch.cuda.device_count()
            if ndev:
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
            # RCCL needs one GPU per rank; fewer GPUs than ranks (a rehearsal box) falls back to gloo
            backend = os.environ.get("AGGFLY_DIST_BACKEND", "nccl" if ndev >= int(os.environ["WORLD_SIZE"]) else "gloo")
            dist.init_process_group(backend)


if __name__ == "__main__":
    cli()
