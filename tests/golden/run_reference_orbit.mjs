// Runs the reference's OrbitCameraAnimator (src/js/animators/OrbitCameraAnimator.js, imported in place through
// esm_loader.mjs with its own gl-matrix module and Ticker) on stand-ins for the camera node, the DOM element, `document`,
// `requestAnimationFrame` and `Date.now`, feeds it scripted input and prints the transforms it assigns, as Float32 bit
// patterns — what Transform's setters store (quat.clone / vec3.clone, Transform.js:98-116).
//   node --experimental-loader ./esm_loader.mjs run_reference_orbit.mjs
import { createRequire } from 'module';
globalThis.glMatrix = createRequire(import.meta.url)('/root/reference/src/lib/gl-matrix-min.js');
globalThis.requestAnimationFrame = () => 0;                       // Ticker.js ticks once at import and never again
globalThis.document = { addEventListener() {} };
let clock = 1000;
Date.now = () => clock;

const bits = a => Array.from(new Uint32Array(new Float32Array(a).buffer));
const element = { addEventListener() {}, removeEventListener() {}, setPointerCapture() {}, releasePointerCapture() {} };
// each step: [op, ...arguments]
const cases = [
    { start: [0, 0, 2], options: {}, steps: [
        ['rotate', 0.3, -0.2], ['rotate', -1.7, 0.9], ['rotate', 0.05, 2.5], ['rotate', -7.0, -4.0], ['zoom', 0.4], ['zoom', -1.3],
        ['move', [0.1, 0, -0.2]], ['rotate', 0.6, 0.1], ['move', [-0.3, 0.05, 0.4]] ] },
    { start: [0.4, -0.3, 1.1], options: { rotationSpeed: 0.01, zoomSpeed: 0.002, moveSpeed: 0.0005 }, steps: [
        ['pointerdown', 0], ['pointermove', 13, -7, false], ['pointermove', -40, 22, true], ['pointerup'], ['pointermove', 5, 5, false],
        ['wheel', 120], ['wheel', -53.5],
        ['keydown', 'w'], ['tick', 16], ['tick', 17], ['keydown', 'D'], ['tick', 33], ['keyup', 'w'], ['tick', 8], ['keyup', 'd'], ['tick', 100],
        ['keydown', 's'], ['keydown', 'a'], ['tick', 21], ['keyup', 's'], ['keyup', 'a'] ] },
    { start: [0, 0, 2], options: {}, steps: [
        ['pointerdown', 1], ['pointermove', 3, 4, false], ['pointerup'], ['rotate', 0.1, 0.1] ] },      // middle-button drag: number * array = NaN, and _move(NaN) throws in strict mode (:88-92,150)
];
async function main() {
const { OrbitCameraAnimator } = await import('/root/reference/src/js/animators/OrbitCameraAnimator.js');
const out = [];
for (const c of cases) {
    const captured = { t: null, r: null };
    const camera = { transform: { get globalTranslation() { return new Float32Array(c.start); }, set localTranslation(v) { captured.t = bits(v); }, set localRotation(v) { captured.r = bits(v); } } };
    clock = 1000;
    const a = new OrbitCameraAnimator(camera, element, c.options);
    const frames = [];
    for (const s of c.steps) {
        captured.t = null; captured.r = null;
        let thrown = null;
        try {
        switch (s[0]) {
            case 'rotate': a._rotateAroundFocus(s[1], s[2]); break;
            case 'zoom': a._zoom(s[1]); break;
            case 'move': a._move(s[1].slice()); break;
            case 'pointerdown': a._handlePointerDown({ button: s[1], pointerId: 1 }); break;
            case 'pointerup': a._handlePointerUp({ pointerId: 1 }); break;
            case 'pointermove': a._handlePointerMove({ movementX: s[1], movementY: s[2], shiftKey: s[3] }); break;
            case 'wheel': a._handleWheel({ deltaY: s[1] }); break;
            case 'keydown': a._handleKeyDown({ key: s[1] }); break;
            case 'keyup': a._handleKeyUp({ key: s[1] }); break;
            case 'tick': clock += s[1]; a._update(); break;
        }
        } catch (e) { thrown = e.constructor.name; }
        frames.push({ step: s, translation_bits: captured.t, rotation_bits: captured.r, throws: thrown });
    }
    out.push({ start: c.start, options: c.options, frames });
}
process.stdout.write(JSON.stringify({ generator: 'tests/golden/run_reference_orbit.mjs', cases: out }));
}
main().catch(e => { console.error(e); process.exit(1); });
