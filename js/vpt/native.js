'use strict';
// Loads the N-API addon (js/addon/vpt_native.node).  There is no JS/CPU fallback for the product path: if the addon
// (or libvpt_hip.so behind it) is missing this throws.
const path = require('path');
let addon = null;
function native() {
    if (!addon) {
        const file = path.join(__dirname, '..', 'addon', 'vpt_native.node');
        try {
            addon = require(file);
        } catch (e) {
            throw new Error('vpt native addon not loadable (' + file + '): ' + e.message +
                ' — build it with `make -C js/addon`; there is no CPU fallback');
        }
    }
    return addon;
}
module.exports = { native };
