import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel
h = Hansel(100, band=2)
print(h.profile_overhead(50))
