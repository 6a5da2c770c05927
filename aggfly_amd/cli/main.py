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
@click.option("--var", default=None, help="Restrict the report to this data variable.")
@click.option("--storage-options", default=None, help="JSON dict for a remote storage backend (accepted; remote stores are not reachable here).")
def info(path, var, storage_options):
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
@click.option("-n", "--rows", default=5, show_default=True, help="Rows to preview. 0 skips the preview.")
@click.option("--uniqueness/--no-uniqueness", default=False, help="Also report which columns are unique across every region.")
@click.option("--regionid", default=None, help="Show the first values of this column.")
@click.option("-v", "--verbose", is_flag=True, help="Show the full traceback on error.")
def regions(path, rows, uniqueness, regionid, verbose):
    """Inspect a region table (attribute table of a shapefile, parquet / feather / csv) to work out its id column."""
    from aggfly_amd.weights import _read_table
    try:
        t = _read_table(path)
    except Exception as e:
        if verbose:
            raise
        raise click.ClickException(f"{type(e).__name__}: {e}")
    click.echo(f"{len(t)} regions; columns: {list(t.columns)}")
    for col in t.columns:
        click.echo(f"  {col}: {t[col].dtype}" + (f"  unique={bool(t[col].is_unique)}" if uniqueness and col != "geometry" else ""))
    if rows:
        click.echo(t.drop(columns=[c for c in ("geometry",) if c in t.columns]).head(rows).to_string())
    if regionid:
        click.echo(f"{regionid}: {t[regionid].head(10).tolist()} ...")


def _load_or_exit(config_path):
    """`load_config` with the reference's error report: every problem of the file, then exit status 1."""
    try:
        return cfg.load_config(config_path)
    except cfg.ConfigError as e:
        click.echo("Config is invalid:", err=True)
        for msg in e.errors:
            click.echo(f"  - {msg}", err=True)
        raise SystemExit(1)


def _resolve_preprocess(c, exit_on_error=False):
    from . import preprocess as pp
    try:
        pp.resolve(c.preprocess, c.preprocess_from)
    except pp.PreprocessError as e:
        if exit_on_error:
            click.echo("Config is invalid:", err=True)
            click.echo(f"  - preprocess: {e}", err=True)
            raise SystemExit(1)
        raise click.ClickException(f"preprocess: {e}")


@cli.command()
@click.argument("config")
@click.option("--strict", is_flag=True, help="Treat unresolved input paths as errors (exit nonzero), not warnings.")
def validate(config, strict):
    """Statically check a config file without reading any data, and print the normalised plan."""
    c = _load_or_exit(config)
    _resolve_preprocess(c, exit_on_error=True)
    warnings = cfg.check_paths(c)
    click.echo(cfg.describe(c))
    if warnings:
        click.echo("")
        click.echo("Errors:" if strict else "Warnings:", err=strict)
        for w in warnings:
            click.echo(f"  - {w}", err=strict)
        if strict:
            raise SystemExit(1)
    click.echo("\nConfig OK.")


@cli.command()
@click.argument("config")
@click.option("--project-dir", default=None, help="Override weights.project_dir (weight cache).")
@click.option("-v", "--verbose", is_flag=True, help="Print per-step progress.")
def weights(config, project_dir, verbose):
    """Locate the precomputed weights a config will use, load them as the run would, and summarise them."""
    c = _load_or_exit(config)
    if project_dir is not None:
        c.project_dir = project_dir
    _resolve_preprocess(c)
    log = click.echo if verbose else (lambda m: None)
    try:
        w, _, _ = pipeline.compute_weights(c, log)
    except Exception as e:
        if verbose:
            raise
        click.echo(f"{type(e).__name__}: {e}", err=True)
        raise SystemExit(1)
    t = w.weights
    click.echo(f"Loaded weights: {len(t)} cell-region rows; {len(t)} (cell, region) pairs; {t['index_right'].nunique()} regions; "
               f"zero_weight={w.zero_weight}")
    if c.project_dir:
        click.echo(f"Cached under: {c.project_dir}")


@cli.command()
@click.argument("config")
@click.option("-o", "--output", default=None, help="Override output.path from the config.")
@click.option("--engine", type=click.Choice(sorted(cfg.ALLOWED_ENGINE)), default=None,
              help="Override the temporal engine (auto/dask/numba/hip: all run the HIP engine).")
@click.option("--years", default=None, help="Override years for a {year}-templated dataset path (e.g. 1980:1990).")
@click.option("--project-dir", default=None, help="Override weights.project_dir (weight cache).")
@click.option("--backend", type=click.Choice(sorted(cfg.ALLOWED_BACKEND)), default=None,
              help="Override execution.backend (accepted; there is no dask cluster here, ranks come from torch.distributed.run).")
@click.option("--n-workers", type=int, default=None, help="Override execution.n_workers (accepted, unused).")
@click.option("-v", "--verbose", is_flag=True, help="Print per-step progress.")
@click.option("--quiet", is_flag=True, hidden=True)
def run(config, output, engine, years, project_dir, backend, n_workers, verbose, quiet):
    """Run the full aggregation pipeline from a config file and write the region-by-period panel."""
    c = _load_or_exit(config)
    if output is not None:
        c.output_path = output
        ext = output.rsplit(".", 1)[-1].lower() if "." in output else ""
        c.output_format = {"pq": "parquet"}.get(ext, ext) or c.output_format
    if engine is not None:
        c.engine = engine
    if project_dir is not None:
        c.project_dir = project_dir
    if backend is not None:
        c.backend = backend
    if n_workers is not None:
        c.n_workers = n_workers
    if years is not None:
        errs = []
        c.years = cfg._parse_years(years, errs)
        if errs:
            raise click.ClickException("; ".join(errs))
    _resolve_preprocess(c)
    _maybe_init_distributed()
    log = click.echo if verbose else (lambda m: None)
    try:
        df = pipeline.run_pipeline(c, log)
    except Exception as e:
        if verbose:
            raise
        raise click.ClickException(f"{type(e).__name__}: {e}")
    from aggfly_amd.distributed import world
    if world()[0] == 0:
        pipeline.write_output(df, c.output_path, c.output_format)
        if not quiet:
            click.echo(f"Wrote {len(df)} rows to {c.output_path} ({c.output_format}).")


def _maybe_init_distributed():
    import os
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            from aggfly_amd.distributed import init_local_rank
            info = init_local_rank(os.environ.get("AGGFLY_DIST_BACKEND"))     # this rank's card + nccl (RCCL) or a gloo rehearsal
            if int(os.environ.get("RANK", "0")) == 0:
                click.echo(f"torch.distributed backend: {info['backend']} ({info['devices_visible']} GPU(s) visible on this node, "
                           f"{os.environ.get('LOCAL_WORLD_SIZE', os.environ['WORLD_SIZE'])} local rank(s))", err=True)


if __name__ == "__main__":
    cli()
