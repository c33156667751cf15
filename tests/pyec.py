"""Independent big-integer affine model of the three curves (test infrastructure).

Deliberately shares nothing with oracle/ (C restatement of the reference) or with the HIP
kernels: textbook affine chord-and-tangent arithmetic on Python ints.  Used as a second opinion
for cases where the reference holds no known-answer vector (variable base, MSM, GLV).
Curve constants: SURVEY.md Appendix A (reference file:line cited there).
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class Curve:
    name: str
    cid: int
    L: int
    p: int
    n: int
    a: int
    b: int
    gx: int
    gy: int
    le: bool = False        # wire records little-endian (bignp256); big-endian everywhere else

    @property
    def order(self):
        return "little" if self.le else "big"


K256 = Curve(
    "k256", 0, 32,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    0, 7,
    0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
    0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8,
)
P256 = Curve(
    "p256", 1, 32,
    0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF,
    0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551,
    -3, 0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B,
    0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
    0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5,
)
P384 = Curve(
    "p384", 2, 48,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFFFF0000000000000000FFFFFFFF,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFC7634D81F4372DDF581A0DB248B0A77AECEC196ACCC52973,
    -3, 0xB3312FA7E23EE7E4988E056BE3F82D19181D9C6EFE8141120314088F5013875AC656398D8A2ED19D2A85C8EDD3EC2AEF,
    0xAA87CA22BE8B05378EB1C71EF320AD746E1D3B628BA79B9859F741E082542A385502F25DBF55296C3A545E3872760AB7,
    0x3617DE4A96262C6F5D9E98BF9292DC29F8F41DBD289A147CE9DA3113B5F0B8C00A60B1CE1D7E819D7A431D7C90EA0E5F,
)
SM2 = Curve(                                                   # sm2/src/arithmetic.rs:53-74, field.rs:34, lib.rs:86
    "sm2", 3, 32,
    0xFFFFFFFEFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF00000000FFFFFFFFFFFFFFFF,
    0xFFFFFFFEFFFFFFFFFFFFFFFFFFFFFFFF7203DF6B21C6052B53BBF40939D54123,
    -3, 0x28E9FA9E9D9F5E344D5A9E4BCF6509A7F39789F515AB8F92DDBCBD414D940E93,
    0x32C4AE2C1F1981195F9904466A39C9948FE30BBFF2660BE1715A4589334C74C7,
    0xBC3736A2F4F6779C59BDCEE36B692153D0A9877CC62A474002DF32E52139F0A0,
)
P224 = Curve(                                                  # p224/src/arithmetic.rs:41-62, field.rs:54-61, lib.rs:50-55
    "p224", 4, 28,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF000000000000000000000001,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFF16A2E0B8F03E13DD29455C5C2A3D,
    -3, 0xB4050A850C04B3ABF54132565044B0B7D7BFD8BA270B39432355FFB4,
    0xB70E0CBD6BB4BF7F321390B94A03C1D356C21122343280D6115C1D21,
    0xBD376388B5F723FB4C22DFE6CD4375A05A07476444D5819985007E34,
)
P192 = Curve(                                                  # p192/src/arithmetic.rs:39-58, field.rs:54, lib.rs:41
    "p192", 5, 24,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFFFFFFFFFFFF,
    0xFFFFFFFFFFFFFFFFFFFFFFFF99DEF836146BC9B1B4D22831,
    -3, 0x64210519E59C80E70FA7E9AB72243049FEB8DEECC146B9B1,
    0x188DA80EB03090F67CBF20EB43A18800F4FF0AFD82FF1012,
    0x07192B95FFC8DA78631011ED6B24CDD573F977A11E794811,
)
P521 = Curve(                                                  # p521/src/arithmetic.rs:57-83, field.rs:68-80, lib.rs:51-60
    "p521", 6, 66,
    0x1FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF,
    0x1FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFA51868783BF2F966B7FCC0148F709A5D03BB5C9B8899C47AEBB6FB71E91386409,
    -3, 0x51953EB9618E1C9A1F929A21A0B68540EEA2DA725B99B315F3B8B489918EF109E156193951EC7E937B1652C0BD3BB1BF073573DF883D2C34F1EF451FD46B503F00,
    0xC6858E06B70404E9CD9E3ECB662395B4429C648139053FB521F828AF606B4D3DBAA14B5E77EFE75928FE1DC127A2FFA8DE3348B3C1856A429BF97E7E31C2E5BD66,
    0x11839296A789A3BC0045C8A5FB42C7D1BD998F54449579B446817AFBD17273E662C97EE72995EF42640C550B9013FAD0761353C7086A272C24088BE94769FD16650,
)
BP256 = Curve(                                                 # bp256/src/r1/arithmetic.rs:34-52, arithmetic/field.rs:53, lib.rs:70
    "bp256", 7, 32,
    0xA9FB57DBA1EEA9BC3E660A909D838D726E3BF623D52620282013481D1F6E5377,
    0xA9FB57DBA1EEA9BC3E660A909D838D718C397AA3B561A6F7901E0E82974856A7,
    0x7D5A0975FC2C3057EEF67530417AFFE7FB8055C126DC5C6CE94A4B44F330B5D9,
    0x26DC5C6CE94A4B44F330B5D9BBD77CBF958416295CF7E1CE6BCCDC18FF8C07B6,
    0x8BD2AEB9CB7E57CB2C4B482FFC81B7AFB9DE27E1E3BD23C23A4453BD9ACE3262,
    0x547EF835C3DAC4FD97F8461A14611DC9C27745132DED8E545C1D54C72F046997,
)
BP384 = Curve(                                                 # bp384/src/r1/arithmetic.rs:32-50, arithmetic/field.rs:53, lib.rs:73
    "bp384", 8, 48,
    0x8CB91E82A3386D280F5D6F7E50E641DF152F7109ED5456B412B1DA197FB71123ACD3A729901D1A71874700133107EC53,
    0x8CB91E82A3386D280F5D6F7E50E641DF152F7109ED5456B31F166E6CAC0425A7CF3AB6AF6B7FC3103B883202E9046565,
    0x7BC382C63D8C150C3C72080ACE05AFA0C2BEA28E4FB22787139165EFBA91F90F8AA5814A503AD4EB04A8C7DD22CE2826,
    0x4A8C7DD22CE28268B39B55416F0447C2FB77DE107DCD2A62E880EA53EEB62D57CB4390295DBC9943AB78696FA504C11,
    0x1D1C64F068CF45FFA2A63A81B7C13F6B8847A3E77EF14FE3DB7FCAFE0CBD10E8E826E03436D646AAEF87B2E247D4AF1E,
    0x8ABE1D7520F9C2A45CB1EB8E95CFD55262B70B29FEEC5864E19C054FF99129280E4646217791811142820341263C5315,
)
BP256T1 = Curve(                                               # bp256/src/t1/arithmetic.rs:35-49 (a = -3), the r1 field and order
    "bp256t1", 9, 32,
    0xA9FB57DBA1EEA9BC3E660A909D838D726E3BF623D52620282013481D1F6E5377,
    0xA9FB57DBA1EEA9BC3E660A909D838D718C397AA3B561A6F7901E0E82974856A7,
    -3, 0x662C61C430D84EA4FE66A7733D0B76B7BF93EBC4AF2F49256AE58101FEE92B04,
    0xA3E8EB3CC1CFE7B7732213B23A656149AFA142C47AAFBC2B79A191562E1305F4,
    0x2D996C823439C56D7F7B22E14644417E69BCB6DE39D027001DABE8F35B25C9BE,
)
BP384T1 = Curve(                                               # bp384/src/t1/arithmetic.rs:35-49 (a = -3), the r1 field and order
    "bp384t1", 10, 48,
    0x8CB91E82A3386D280F5D6F7E50E641DF152F7109ED5456B412B1DA197FB71123ACD3A729901D1A71874700133107EC53,
    0x8CB91E82A3386D280F5D6F7E50E641DF152F7109ED5456B31F166E6CAC0425A7CF3AB6AF6B7FC3103B883202E9046565,
    -3, 0x7F519EADA7BDA81BD826DBA647910F8C4B9346ED8CCDC64E4B1ABD11756DCE1D2074AA263B88805CED70355A33B471EE,
    0x18DE98B02DB9A306F2AFCD7235F72A819B80AB12EBD653172476FECD462AABFFC4FF191B946A5F54D8D0AA2F418808CC,
    0x25AB056962D30651A114AFD2755AD336747F93475B7A1FCA3B88F2B6A208CCFE469408584DC2B2912675BF5B9E582928,
)
BIGN256 = Curve(                                               # bignp256/src/arithmetic.rs:39-53, arithmetic/field.rs:60-66, lib.rs:74,102
    "bign256", 11, 32,
    2 ** 256 - 189,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFD95C8ED60DFB4DFC7E5ABF99263D6607,
    -3, 0x77CE6C1515F3A8EDD2C13AABE4D8FBBE4CF55069978B9253B22E7D6BD69C03F1,
    0,
    0x6BF7FC3CFB16D69F5CE4C9A351D6835D78913966C408F6521E29CF1804516A93,
    le=True,
)
CURVES = {"k256": K256, "p256": P256, "p384": P384, "sm2": SM2, "p224": P224, "p192": P192, "p521": P521, "bp256": BP256, "bp384": BP384, "bp256t1": BP256T1, "bp384t1": BP384T1, "bign256": BIGN256}

# secp256k1 endomorphism constants (k256/src/arithmetic/mul.rs:4-5, projective.rs:31-37)
K256_LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
K256_BETA = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE

INF = None  # point at infinity


def on_curve(c, P):
    if P is INF:
        return True
    x, y = P
    return 0 <= x < c.p and 0 <= y < c.p and (y * y - (x * x * x + c.a * x + c.b)) % c.p == 0


def neg(c, P):
    if P is INF:
        return INF
    return (P[0], (-P[1]) % c.p)


def add(c, P, Q):
    if P is INF:
        return Q
    if Q is INF:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % c.p == 0:
            return INF
        lam = (3 * x1 * x1 + c.a) * pow(2 * y1, -1, c.p) % c.p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, c.p) % c.p
    x3 = (lam * lam - x1 - x2) % c.p
    return (x3, (lam * (x1 - x3) - y1) % c.p)


def mul(c, k, P):
    k %= c.n
    R = INF
    Q = P
    while k:
        if k & 1:
            R = add(c, R, Q)
        Q = add(c, Q, Q)
        k >>= 1
    return R


def G(c):
    return (c.gx, c.gy)


def msm(c, ks, Ps):
    R = INF
    for k, P in zip(ks, Ps):
        R = add(c, R, mul(c, k, P))
    return R


# ---- wire-format helpers (fixed width, big-endian unless the curve says `le`; identity = zeros + flag) ----

def enc_scalar(c, k):
    return int(k).to_bytes(c.L, c.order)


def enc_point(c, P):
    """-> (xy bytes, inf flag)"""
    if P is INF:
        return bytes(2 * c.L), 1
    return P[0].to_bytes(c.L, c.order) + P[1].to_bytes(c.L, c.order), 0


def dec_point(c, xy, inf):
    if inf:
        assert xy == bytes(2 * c.L), "identity must be encoded as zeros"
        return INF
    return (int.from_bytes(xy[: c.L], c.order), int.from_bytes(xy[c.L:], c.order))


# ---- ECDSA (SEC1 v2 4.1.3 / 4.1.4), the independent model for the verification tests --------------------------

def ecdsa_sign(c, d, z, k):
    """(r, s) for private key d, digest integer z, nonce k (no low-S normalisation)."""
    R = mul(c, k, G(c))
    r = R[0] % c.n
    s = pow(k, -1, c.n) * (z + r * d) % c.n
    return r, s


def ecdsa_verify(c, Q, z, r, s, reject_high_s=False):
    if not (1 <= r < c.n and 1 <= s < c.n):
        return False
    if reject_high_s and s > (c.n - 1) // 2:
        return False
    if Q is INF or not on_curve(c, Q):
        return False
    w = pow(s, -1, c.n)
    R = add(c, mul(c, z % c.n * w % c.n, G(c)), mul(c, r * w % c.n, Q))
    return R is not INF and R[0] % c.n == r


def sqrt_mod(a, p):
    """A square root of a modulo the odd prime p, or None: one exponentiation for p = 3 (mod 4), otherwise Tonelli-Shanks
    in its discrete-logarithm form (p224: p - 1 = 2^96 (2^128 - 1))."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    s, q = 0, p - 1
    while q % 2 == 0:
        s, q = s + 1, q // 2
    z = next(c for c in range(2, 200) if pow(c, (p - 1) // 2, p) == p - 1)
    g, t, e = pow(z, q, p), pow(a, q, p), 0
    for i in range(1, s):                       # e: t = g^e, e even; bit i from (t g^-e)^(2^(s-1-i))
        if pow(t * pow(g, -e, p) % p, 1 << (s - 1 - i), p) != 1:
            e |= 1 << i
    y = pow(a, (q + 1) // 2, p) * pow(g, -(e // 2), p) % p
    assert y * y % p == a
    return y


def lift_x(c, x, y_is_odd):
    """The curve point with this x and the requested y parity, or None (x >= p, or x^3 + a x + b is not a square)."""
    if x >= c.p:
        return None
    alpha = (pow(x, 3, c.p) + c.a * x + c.b) % c.p
    y = sqrt_mod(alpha, c.p)
    if y is None:
        return None
    if (y & 1) != int(bool(y_is_odd)):
        y = (c.p - y) % c.p
    return (x, y)


def ecdsa_recover(c, z, r, s, recid, reject_high_s=False):
    """The public key (affine) a signature recovers to under a recovery id byte (bit 0: y(R) odd, bit 1: x(R) = r + n),
    or None — SEC1 v2 4.1.6 for one candidate, followed by the verification the `ecdsa` crate runs on the result."""
    if recid > 3 or not (1 <= r < c.n and 1 <= s < c.n):
        return None
    x = r + (c.n if recid & 2 else 0)
    if x >= 1 << (8 * c.L):
        return None
    R = lift_x(c, x, recid & 1)
    if R is None:
        return None
    rinv = pow(r, -1, c.n)
    Q = add(c, mul(c, (-rinv * (z % c.n)) % c.n, G(c)), mul(c, rinv * s % c.n, R))
    if Q is INF or not ecdsa_verify(c, Q, z, r, s, reject_high_s):
        return None
    return Q


def keccak256(data):
    """Keccak-256 (the pre-standard padding 0x01 that Ethereum uses — hashlib's sha3_256 pads with 0x06)."""
    RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
          0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
          0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
          0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
          0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
    ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
    M = (1 << 64) - 1
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & M if n else v
    rate = 136
    msg = bytearray(data) + b"\x01" + bytes((-len(data) - 2) % rate) + b"\x80" if (len(data) + 1) % rate else bytearray(data) + b"\x81"
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
        for rc in RC:
            C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
            D = [C[(x - 1) % 5] ^ rol(C[(x + 1) % 5], 1) for x in range(5)]
            A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
            B = [[0] * 5 for _ in range(5)]
            for x in range(5):
                for y in range(5):
                    B[y][(2 * x + 3 * y) % 5] = rol(A[x][y], ROT[x][y])
            A = [[B[x][y] ^ (~B[(x + 1) % 5][y] & M & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
            A[0][0] ^= rc
    return b"".join(A[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


# ---- SM2DSA (GB/T 32918.2 / draft-shen-sm2-ecdsa-02 5.2-5.3), the independent model ---------------------------------------

def sm2dsa_sign(c, d, e, k):
    """(r, s) for private key d, digest integer e (= SM3(ZA || M)), nonce k; None if the nonce has to be rejected."""
    x1 = mul(c, k, G(c))[0]
    r = (e + x1) % c.n
    if r == 0 or r + k == c.n:
        return None
    s = pow(1 + d, -1, c.n) * (k - r * d) % c.n
    return (r, s) if s else None


def sm2dsa_verify(c, Q, e, r, s):
    if not (1 <= r < c.n and 1 <= s < c.n):
        return False
    t = (r + s) % c.n
    if t == 0 or Q is INF or not on_curve(c, Q):
        return False
    R = add(c, mul(c, s, G(c)), mul(c, t, Q))
    x1 = 0 if R is INF else R[0]
    return (e % c.n + x1) % c.n == r


def sm2_za(c, ident, Q):
    """ZA = SM3(ENTL || ID || a || b || xG || yG || xA || yA) (draft-shen-sm2-ecdsa-02 5.1.4.4; sm2/src/dsa.rs hash_z)."""
    import hashlib
    f = lambda v: (v % c.p).to_bytes(32, "big")
    data = (8 * len(ident)).to_bytes(2, "big") + ident + f(c.a) + f(c.b) + f(c.gx) + f(c.gy) + f(Q[0]) + f(Q[1])
    return hashlib.new("sm3", data).digest()


# ---- belt-hash (STB 34.101.31-2020 §7.8) and bign signatures (STB 34.101.45-2013 §7), the independent model -----------------

BELT_H = bytes.fromhex(
    "B194BAC80A08F53B366D008E584A5DE48504FA9D1BB6C7AC252E72C202FDCE0D5BE3D61217B96181FE6786AD716B890B5CB0C0FF33C356B835C405AED8E07F99"
    "E12BDC1AE28257EC703FCCF095EE8DF1C1AB76389FE678CAF7C6F860D5BB9C4FF33C657B637C306ADD4EA7799EB23D313E98B56E27D3BCCF591E181F4C5AB793"
    "E9DEE72C8F0C0FA62DDB49F46F73964706075316ED247A3739CBA38303A98BF692BD9B1CE5D141015445FBC95E4D0EF2682080AA227D642F2687F93490405511"
    "BE32971343FC9A48A02A885F194B09A17ECDA4D01544AF8CA58450BF66D2E88AA2D7465242A8DFB36974C551EB232921D4EFD9B43A622875911410EA776CDA1D")
BELT_OID = bytes([0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x1F, 0x51])     # bignp256/src/ecdsa.rs:58-60
assert len(BELT_H) == 256 and sorted(BELT_H) == list(range(256))                          # the S-box is a permutation
_M32 = 0xFFFFFFFF


def _belt_g(u, r):
    v = BELT_H[u & 255] | BELT_H[(u >> 8) & 255] << 8 | BELT_H[(u >> 16) & 255] << 16 | BELT_H[(u >> 24) & 255] << 24
    return ((v << r) | (v >> (32 - r))) & _M32


def belt_block(x, key):
    """belt-block encryption of the 16-byte block x under the 32-byte key (§7.1.3)."""
    a, b, c, d = (int.from_bytes(x[4 * i:4 * i + 4], "little") for i in range(4))
    k = [int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)]
    K = lambda j: k[(j - 1) % 8]
    for i in range(1, 9):
        b ^= _belt_g((a + K(7 * i - 6)) & _M32, 5)
        c ^= _belt_g((d + K(7 * i - 5)) & _M32, 21)
        a = (a - _belt_g((b + K(7 * i - 4)) & _M32, 13)) & _M32
        e = _belt_g((b + c + K(7 * i - 3)) & _M32, 21) ^ i
        b = (b + e) & _M32
        c = (c - e) & _M32
        d = (d + _belt_g((c + K(7 * i - 2)) & _M32, 13)) & _M32
        b ^= _belt_g((a + K(7 * i - 1)) & _M32, 21)
        c ^= _belt_g((d + K(7 * i)) & _M32, 5)
        a, b = b, a
        c, d = d, c
        b, c = c, b
    return b"".join(v.to_bytes(4, "little") for v in (b, d, a, c))


def _bxor(p, q):
    return bytes(i ^ j for i, j in zip(p, q))


def _belt_sigma1(u):
    t = _bxor(u[32:48], u[48:64])
    return _bxor(belt_block(t, u[0:32]), t)


def _belt_sigma2(u):
    s1 = _belt_sigma1(u)
    return (_bxor(belt_block(u[0:16], s1 + u[48:64]), u[0:16]) +
            _bxor(belt_block(u[16:32], _bxor(s1, b"\xff" * 16) + u[32:48]), u[16:32]))


def belt_hash(msg):
    s, h = bytes(16), BELT_H[:32]
    for off in range(0, len(msg), 32):
        x = msg[off:off + 32].ljust(32, b"\0")
        s = _bxor(s, _belt_sigma1(x + h))
        h = _belt_sigma2(x + h)
    return _belt_sigma2((8 * len(msg)).to_bytes(16, "little") + s + h)


def bign_sign(c, d, h, k):
    """48-byte signature S0 || S1 of the 32-byte hash h under private key d with nonce k (STB 34.101.45 §7.1 steps 4-8)."""
    R = mul(c, k, G(c))
    s0 = belt_hash(BELT_OID + R[0].to_bytes(32, "little") + h)[:16]
    s1 = (k - int.from_bytes(h, "little") - (int.from_bytes(s0, "little") + 2 ** 128) * d) % c.n
    return s0 + s1.to_bytes(32, "little")


def bign_verify(c, Q, h, sig):
    """bignp256/src/ecdsa/verifying.rs:100-147 with Signature::from_bytes (ecdsa.rs:72-88)."""
    s0, s1 = int.from_bytes(sig[:16], "little"), int.from_bytes(sig[16:48], "little")
    if s0 == 0 or s1 == 0 or s1 >= c.n or Q is INF or Q[0] >= c.p or Q[1] >= c.p or not on_curve(c, Q):
        return False
    R = add(c, mul(c, (s1 + int.from_bytes(h, "little")) % c.n, G(c)), mul(c, (s0 + 2 ** 128) % c.n, Q))
    if R is INF:
        return False
    return belt_hash(BELT_OID + R[0].to_bytes(32, "little") + h)[:16] == sig[:16]
