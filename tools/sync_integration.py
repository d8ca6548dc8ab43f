"""Re-embed examples/ctypes_stub.py into INTEGRATION.md section 2 (between the BEGIN / END markers), so that the document shows the
file verbatim; tests/test_cabi_cpu.py::test_documented_ctypes_stub_matches_the_library fails when the two drift apart."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
doc_path = os.path.join(ROOT, "INTEGRATION.md")
doc = open(doc_path).read()
src = open(os.path.join(ROOT, "examples", "ctypes_stub.py")).read().strip()
block = "<!-- BEGIN examples/ctypes_stub.py -->\n```python\n" + src + "\n```\n<!-- END examples/ctypes_stub.py -->"
new, n = re.subn(r"<!-- BEGIN examples/ctypes_stub.py -->.*?<!-- END examples/ctypes_stub.py -->", lambda m: block, doc, flags=re.S)
assert n == 1, "markers not found in INTEGRATION.md"
open(doc_path, "w").write(new)
print("INTEGRATION.md section 2 synchronised with examples/ctypes_stub.py")
