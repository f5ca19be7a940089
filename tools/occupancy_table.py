"""Occupancy / round quantisation of every launch shape in a rocprofv3 kernel trace (rocpd sqlite): workgroups per launch against the
resident slots (256 CUs x workgroups per CU from LDS bytes, VGPRs and wave count), average duration.  A launch with 1.04 rounds pays two.
usage: python tools/occupancy_table.py results.db [name-filter-regex]"""
import re, sqlite3, sys

def short(n):
    n = re.sub(r"^void ", "", n).replace("dgsct::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:58]

def main(path, flt=None):
    c = sqlite3.connect(path)
    q = ("select name, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, lds_size, vgpr_count, accum_vgpr_count, count(*), avg(end-start) "
         "from kernels group by 1,2,3,4,5,6,7,8,9,10 order by count(*)*avg(end-start) desc")
    print(f"{'calls':>6} {'avg us':>8} {'WGs':>7} {'LDS KB':>7} {'VGPR':>5} {'WG/CU':>6} {'rounds':>7}  kernel")
    for name, gx, gy, gz, wx, wy, wz, lds, vg, ag, n, avg in c.execute(q):
        if flt and not re.search(flt, name): continue
        wg = (gx // max(wx, 1)) * (gy // max(wy, 1)) * (gz // max(wz, 1))
        waves = (wx * wy * wz + 63) // 64
        regs = max(vg + ag, 1)
        wps = max(1, min(8, 512 // (((regs + 7) // 8) * 8)))          # waves per SIMD by registers (512 per lane on gfx950)
        by_reg = (wps * 4) // waves if waves <= wps * 4 else 0
        by_lds = (160 * 1024) // lds if lds else 99
        by_wave = 32 // waves
        per_cu = max(1, min(by_reg if by_reg else 1, by_lds, by_wave))
        print(f"{n:6d} {avg/1e3:8.1f} {wg:7d} {lds/1024:7.1f} {regs:5d} {per_cu:6d} {wg/(256*per_cu):7.2f}  {short(name)}")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
