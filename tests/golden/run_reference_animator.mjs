// Runs the reference's CircleAnimator (src/js/animators/CircleAnimator.js, imported in place through esm_loader.mjs, with
// its own gl-matrix module) on a stand-in node and prints the transforms it assigns, as Float32 bit patterns — what
// Transform's setters store (quat.clone / vec3.clone, Transform.js:98-116).
//   node --experimental-loader ./esm_loader.mjs run_reference_animator.mjs
import { createRequire } from 'module';
// the reference's gl-matrix-module.js re-exports the global its UMD build defines in the browser
globalThis.glMatrix = createRequire(import.meta.url)('/root/reference/src/lib/gl-matrix-min.js');

const bits = a => Array.from(new Uint32Array(new Float32Array(a).buffer));
const cases = [
    { options: {}, times: [0, 0.125, 0.3, 0.77, 1.5] },
    { options: { center: [0, 0, 2], direction: [0, 0, 1], radius: 0.01, frequency: 1 }, times: [0, 0.1, 0.45] },       // RenderingContext.js:48-53
    { options: { center: [0.1, -0.2, 0.3], direction: [1, 2, 0.5], radius: 1.8, frequency: 0.37 }, times: [0, 0.2, 1.1, 7.9] },
    { options: { center: [0, 0, 0], direction: [0, 1, 0], radius: 2.0, frequency: 0.05 }, times: [0, 1, 2, 3, 10] },
];
async function main() {
const { CircleAnimator } = await import('/root/reference/src/js/animators/CircleAnimator.js');
const out = [];
for (const c of cases) {
    const captured = {};
    const node = { transform: { set localTranslation(v) { captured.t = bits(v); }, set localRotation(v) { captured.r = bits(v); } } };
    const anim = new CircleAnimator(node, c.options);
    const frames = [];
    for (const t of c.times) { anim.update(t); frames.push({ t, translation_bits: captured.t, rotation_bits: captured.r }); }
    out.push({ options: c.options, frames });
}
process.stdout.write(JSON.stringify({ generator: 'tests/golden/run_reference_animator.mjs', cases: out }));
}
main().catch(e => { console.error(e); process.exit(1); });
