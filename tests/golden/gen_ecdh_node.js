// Generates tests/golden/ecdh_node.json: ECDH shared secrets computed by Node's crypto module (OpenSSL) — an
// implementation independent of both the reference and this repository (SURVEY.md §8c "third opinion").
//     node tests/golden/gen_ecdh_node.js > tests/golden/ecdh_node.json
const crypto = require('crypto');
const curves = {k256: 'secp256k1', p256: 'prime256v1', p384: 'secp384r1', sm2: 'SM2', p224: 'secp224r1', p192: 'prime192v1', p521: 'secp521r1', bp256: 'brainpoolP256r1', bp384: 'brainpoolP384r1', bp256t1: 'brainpoolP256t1', bp384t1: 'brainpoolP384t1'};
const out = {source: 'node ' + process.version + ' crypto.createECDH (OpenSSL ' + process.versions.openssl + ')'};
for (const [name, ossl] of Object.entries(curves)) {
  const rows = [];
  for (let i = 0; i < 24; i++) {
    const a = crypto.createECDH(ossl), b = crypto.createECDH(ossl);
    a.generateKeys(); b.generateKeys();
    const pub = b.getPublicKey(null, 'uncompressed');           // 04 || x || y
    const own = a.getPublicKey(null, 'uncompressed');          // d * G: a fixed-base anchor as well
    rows.push({d: a.getPrivateKey('hex').padStart((pub.length - 1), '0'), px: own.slice(1, 1 + (own.length - 1) / 2).toString('hex'),
               py: own.slice(1 + (own.length - 1) / 2).toString('hex'), qx: pub.slice(1, 1 + (pub.length - 1) / 2).toString('hex'),
               qy: pub.slice(1 + (pub.length - 1) / 2).toString('hex'), z: a.computeSecret(pub).toString('hex')});
  }
  out[name] = rows;
}
console.log(JSON.stringify(out, null, 1));
