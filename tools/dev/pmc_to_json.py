"""profiles/pmc_traffic.json (what bench.py's roofline.traffic reads) from a pmc_hbm_traffic.csv of tools/dev/collect_profiles.sh:
pmc_to_json.py <pmc_hbm_traffic.csv> <profiles/pmc_traffic.json>   (keeps the file's `source` text)"""
import csv, json, sys
dst = json.load(open(sys.argv[2]))
fam = {}
for r in csv.DictReader(open(sys.argv[1])):
    fam[r["family"]] = {"launches": int(r["launches"]), "hbm_bytes_per_step": float(r["total_hbm_bytes_per_step"])}
dst["families"] = fam
json.dump(dst, open(sys.argv[2], "w"), indent=1)
print({k: v["hbm_bytes_per_step"] for k, v in fam.items()})
