"""Tiny TOML writer for flat {section: {key: scalar}} dicts (golden generation only)."""


def _fmt(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, float)):
        return repr(v)
    return '"' + str(v).replace("\\", "\\\\").replace('"', '\\"') + '"'


def dump(d, fh):
    out = []
    for sec, kv in d.items():
        out.append(f"[{sec}]")
        for k, v in kv.items():
            out.append(f"{k} = {_fmt(v)}")
        out.append("")
    fh.write("\n".join(out).encode())
