"""Merges the per-ray counter records of several tools/profile_workloads.sh runs (reference trees, SAH
trees, the opt-in walks) into one file:  python tools/merge_pmc.py <out.json> <in.json> ...
Later files win where two hold the same key."""
import json
import sys


def main():
    out = {"note": None, "records": {}}
    for path in sys.argv[2:]:
        with open(path) as f:
            d = json.load(f)
        out["note"] = out["note"] or d.get("note")
        out["records"].update(d["records"])
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(f"{len(out['records'])} records:", ", ".join(out["records"]))


if __name__ == "__main__":
    main()
