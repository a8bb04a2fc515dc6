"""Hash of the library's device and host sources with comments and white space removed: ties profiles/traffic.json (PMC passes of one
build) to the tree bench.py runs from.  A change of a comment is not a change of the build."""
import hashlib
import re
from pathlib import Path

_TOKEN = re.compile(r'"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'|//[^\n]*|/\*.*?\*/', re.S)


def strip_comments(text: str) -> str:
    text = _TOKEN.sub(lambda m: m.group(0) if m.group(0)[0] in "\"'" else " ", text)
    return " ".join(text.split())


def code_hash(root: Path) -> str:
    h = hashlib.sha256()
    for f in sorted((Path(root) / "elastic_elgamal_amd" / "csrc").iterdir()):
        if f.suffix in (".cuh", ".hip", ".hpp", ".h"):
            h.update(f.name.encode())
            h.update(strip_comments(f.read_text()).encode())
    return h.hexdigest()[:16]
