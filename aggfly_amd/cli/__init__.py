"""YAML-driven command line on top of the HIP engine (`aggfly {info,regions,validate,weights,run}`).

Keeps the reference CLI's config schema (`aggfly/cli/config.py:214-386`, example
`examples/era5_counties_area.yaml`) so an existing config runs unchanged apart from where
regions and weights come from: this engine consumes the region TABLE and the precomputed
weights TABLE / cached ``.feather`` instead of recomputing geometry.
"""
