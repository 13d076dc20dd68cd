'use strict';
// src/js/tonemappers/RangeToneMapper.js:10-66
const { AbstractToneMapper, P } = require('./AbstractToneMapper.js');
const { native } = require('../native.js');

class RangeToneMapper extends AbstractToneMapper {

static KIND() { return native().VPT_TONEMAPPER_RANGE; }

constructor(gl, texture, options) {
    super(gl, texture, options);
    this.registerProperties([                                                               // :14-34
        { name: 'min', label: 'Min', type: 'spinner', value: 0 },
        { name: 'max', label: 'Max', type: 'spinner', value: 1 },
        { name: 'gamma', label: 'Gamma', type: 'spinner', value: 2.2, min: 0 },
    ]);
}

_params() {                                                                                 // :58-60
    const p = super._params();
    p[P.MIN] = this.min; p[P.MAX] = this.max; p[P.GAMMA] = this.gamma;
    return p;
}

}
module.exports = { RangeToneMapper };
