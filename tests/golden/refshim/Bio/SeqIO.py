class _Rec:
    def __init__(self, name, seq):
        self.id = self.name = name
        self.description = name
        self.seq = seq


def parse(handle, fmt="fasta"):
    assert fmt == "fasta"
    own = False
    if isinstance(handle, (str, bytes)) or hasattr(handle, "__fspath__"):
        handle = open(handle, "r")
        own = True
    try:
        name, chunks = None, []
        for line in handle:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    yield _Rec(name, "".join(chunks))
                name, chunks = line[1:].split(" ")[0] if len(line) > 1 else "", []
            elif name is not None:
                chunks.append("".join(line.split()))
        if name is not None:
            yield _Rec(name, "".join(chunks))
    finally:
        if own:
            handle.close()
