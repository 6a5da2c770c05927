import sys, os, cProfile, pstats
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import torch, numpy as np, pandas as pd
import aggfly_amd as af
from aggfly_amd import synth
T, ny, nx, R = 91615, 180, 288, 3600
g = torch.Generator(device="cuda").manual_seed(4)
cube = (14 + 12 * torch.randn((T, ny, nx), generator=g, device="cuda", dtype=torch.float32))
tindex = af.cf_range("1850-01-01", T, "D", "noleap")
lat, lon = -89.5 + 1.0 * np.arange(ny), 0.625 + 1.25 * np.arange(nx)
ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"], {"time": tindex, "latitude": lat, "longitude": lon}), lon_is_360=True)
tab = synth.weights_table(ny, nx, R, seed=5)
gr = af.GeoRegions(pd.DataFrame({"gid": [f"g{i:05d}" for i in range(int(tab.index_right.max()) + 1)]}), regionid="gid")
w = af.weights_from_objects(ds, gr, table=tab)
edges = np.arange(-20, 50, 5.0)
spec = dict(bins=[("aggregate", {"calc": "bins", "groupby": "year", "ddargs": [[edges[i], edges[i + 1], 0] for i in range(13)]})])
af.aggregate_dataset(dataset=ds, weights=w, **spec)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): df = af.aggregate_dataset(dataset=ds, weights=w, **spec); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
