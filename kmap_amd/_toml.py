"""TOML load/dump for config.toml (py3.10: no tomllib; tomli_w is not installed)."""
try:
    import tomllib as _toml_reader
except ModuleNotFoundError:  # Python < 3.11
    import tomli as _toml_reader


def load_toml(path):
    with open(path, "rb") as fh:
        return _toml_reader.load(fh)


def _scalar(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, float)):
        return repr(v)
    if isinstance(v, str):
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(_scalar(x) for x in v) + "]"
    raise TypeError(f"cannot write {type(v)} to TOML")


def dump_toml(cfg, path):
    """Writes {section: {key: scalar}} in the layout tomli_w produces (one blank line between tables)."""
    lines = []
    for section, table in cfg.items():
        lines.append(f"[{section}]")
        lines.extend(f"{k} = {_scalar(v)}" for k, v in table.items())
        lines.append("")
    with open(path, "w") as fh:
        fh.write("\n".join(lines))
