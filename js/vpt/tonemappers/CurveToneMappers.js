'use strict';
// src/js/tonemappers/{Reinhard,Reinhard2,Uncharted2,Filmic,Unreal,Aces,Lottes,Uchimura}ToneMapper.js — eight classes whose
// host side differs only in the shader they build (each file :10-59); here only the native kind differs.
const { ExposureGammaToneMapper } = require('./AbstractToneMapper.js');
const { native } = require('../native.js');

class ReinhardToneMapper extends ExposureGammaToneMapper { static KIND() { return native().VPT_TONEMAPPER_REINHARD; } }
class Reinhard2ToneMapper extends ExposureGammaToneMapper { static KIND() { return native().VPT_TONEMAPPER_REINHARD2; } }
class Uncharted2ToneMapper extends ExposureGammaToneMapper { static KIND() { return native().VPT_TONEMAPPER_UNCHARTED2; } }
class FilmicToneMapper extends ExposureGammaToneMapper { static KIND() { return native().VPT_TONEMAPPER_FILMIC; } }
class UnrealToneMapper extends ExposureGammaToneMapper { static KIND() { return native().VPT_TONEMAPPER_UNREAL; } }
class AcesToneMapper extends ExposureGammaToneMapper { static KIND() { return native().VPT_TONEMAPPER_ACES; } }
class LottesToneMapper extends ExposureGammaToneMapper { static KIND() { return native().VPT_TONEMAPPER_LOTTES; } }
class UchimuraToneMapper extends ExposureGammaToneMapper { static KIND() { return native().VPT_TONEMAPPER_UCHIMURA; } }

module.exports = { ReinhardToneMapper, Reinhard2ToneMapper, Uncharted2ToneMapper, FilmicToneMapper, UnrealToneMapper,
                   AcesToneMapper, LottesToneMapper, UchimuraToneMapper };
