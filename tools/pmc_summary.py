"""Reduce rocprofv3 --pmc counter CSVs to per-kernel means and apply the gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB, and FETCH_SIZE reports
exactly half of the bytes of a 16-byte-per-lane coalesced streaming read, so it is doubled."""
import csv, collections, json, sys
out = {}
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void l2k::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, vals in cs.items():
            vals = vals[2:] if len(vals) > 4 else vals      # drop the warm-up launches
            out.setdefault(k, {})[c] = sum(vals) / len(vals)
for k, cs in out.items():
    if "FETCH_SIZE" in cs:
        cs["hbm_read_bytes_corrected"] = cs["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in cs:
        cs["hbm_write_bytes"] = cs["WRITE_SIZE"] * 1024
print(json.dumps(out, indent=1))
