"""Extract oxDNA's literal model constants into a JSON fixture.

Source: <reference>/data/templates/model_template.h (the ``#define NAME value`` rows of the
oxDNA ``model.h`` template the reference ships).  Only NAME -> numeric value pairs are kept
(data, not source text).  Run in the build container where /root/reference exists:

    python tests/golden/make_model_constants.py
"""

import json
import math
import re
from pathlib import Path

SRC = Path("/root/reference/data/templates/model_template.h")
DST = Path(__file__).resolve().parent / "oxdna_model_constants.json"


def evaluate(expr: str) -> float | None:
    expr = re.sub(r"(?<=[0-9.])f\b", "", expr).replace("PI", repr(math.pi))
    if not re.fullmatch(r"[0-9eE+\-*/(). ]+", expr):
        return None
    try:
        return float(eval(expr, {"__builtins__": {}}))  # noqa: S307 - digits and operators only
    except Exception:  # noqa: BLE001
        return None


def main() -> None:
    out = {}
    for line in SRC.read_text().splitlines():
        m = re.match(r"#define\s+(\w+)\s+(.+?)\s*(//.*)?$", line)
        if not m:
            continue
        v = evaluate(m.group(2).strip())
        if v is not None:
            out[m.group(1)] = v
    DST.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print(f"wrote {len(out)} constants to {DST}")


if __name__ == "__main__":
    main()
