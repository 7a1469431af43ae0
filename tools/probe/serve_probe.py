"""drives tools/probe/serve_probe (euclid_serve on the GPU) with Python integers as the client, like
tests/test_hostsim_device_code.py::_serve_sequence does with the host simulator.  usage: serve_probe.py <binary> <ops file>
ops file lines: "<stop_bits> <x hex> <y hex>" """
import subprocess, sys, math

def limbs(v):
    return " ".join("%x" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(40))

def run(binary, x, y, stop):
    pr = subprocess.Popen([binary], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    tx, ty, sd = 39, 39, 0
    ux, uy = 0, 1
    rounds = 0
    x0, y0 = x, y
    try:
        while True:
            pr.stdin.write("%d %d %d %d %s %s\n" % (stop, tx, ty, sd, limbs(x), limbs(y)))
            pr.stdin.flush()
            out = pr.stdout.readline().split()
            tx, ty, sd = int(out[0]), int(out[1]), int(out[2])
            w = [int(v, 16) for v in out[3:]]
            A, ok = w[0] & 0x7FFFFFFF, w[0] >> 31
            B, dn = w[1] & 0x7FFFFFFF, w[1] >> 31
            Cc, D = w[2], w[3]
            if dn:
                break
            rounds += 1
            if rounds > 400:
                return "NO END after 400 rounds"
            if ok:
                nx, ny = A * x - B * y, D * y - Cc * x
                if nx < 0 or ny < 0 or not (B | Cc):
                    return "round %d: INVALID matrix %s for x bits %d y bits %d (tx %d ty %d)" % (rounds, (A, B, Cc, D), x.bit_length(), y.bit_length(), tx, ty)
                ux, uy = A * ux + B * uy, D * uy + Cc * ux
                x, y = nx, ny
            else:
                if x < y:
                    x, y, ux, uy = y, x, uy, ux
                q = x // y
                cut = max(0, q.bit_length() - 32)
                q = (q >> cut) << cut
                x, ux = x - q * y, ux + q * uy
            if not (x < (1 << (32 * (tx + 1))) and y < (1 << (32 * (ty + 1)))):
                return "round %d: STALE hints tx %d ty %d for x bits %d y bits %d" % (rounds, tx, ty, x.bit_length(), y.bit_length())
    finally:
        pr.stdin.close()
        pr.wait()
    if stop < 0:
        return "ok, %d rounds, gcd %s" % (rounds, "correct" if max(x, y) == math.gcd(x0, y0) else "WRONG")
    return "ok, %d rounds, final bits %d / %d (stop %d)" % (rounds, x.bit_length(), y.bit_length(), stop)

if __name__ == "__main__":
    for line in open(sys.argv[2]):
        p = line.split()
        if len(p) == 3:
            print(p[0], run(sys.argv[1], int(p[1], 16), int(p[2], 16), int(p[0])))
