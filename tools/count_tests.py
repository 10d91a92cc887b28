"""Counts the collected tests per marker (what README.md / DESIGN.md quote): python tools/count_tests.py [--write]
--write replaces the text between <!-- TESTCOUNT --> markers in README.md."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def count(marker):
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "--collect-only", "-q", "-m", marker], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    m = re.search(r"(\d+)(?:/\d+)? tests? (?:collected|selected)|(\d+) selected", r.stdout)
    m2 = re.search(r"(\d+)/(\d+) tests collected \((\d+) deselected\)", r.stdout)
    if m2:
        return int(m2.group(1))
    m3 = re.search(r"(\d+) tests? collected", r.stdout)
    return int(m3.group(1)) if m3 else -1
gpu, cpu = count("gpu"), count("not gpu")
text = "%d GPU parity tests (`pytest -m gpu`), %d CPU tests (`pytest -m \"not gpu\"`)" % (gpu, cpu)
print(text)
if "--write" in sys.argv:
    p = os.path.join(ROOT, "README.md")
    s = open(p).read()
    s2 = re.sub(r"<!-- TESTCOUNT -->.*?<!-- /TESTCOUNT -->", "<!-- TESTCOUNT -->" + text + "<!-- /TESTCOUNT -->", s, flags=re.S)
    open(p, "w").write(s2)
