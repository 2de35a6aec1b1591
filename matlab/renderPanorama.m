function [panorama, rgbAnnotation] = renderPanorama(input, images, imgSize, cameras, mode, refIdx, opts)
    %RENDERPANORAMA Shadows PP/renderPanorama/renderPanorama.m (same signature, drop-in on the MATLAB path).
    %   The host geometry (option defaults :41-71, auto reference :84-122, bounds :1507-1754, canvas size :125-232)
    %   is evaluated here in double exactly as the reference does; the tile loop (:342-425: rays, sampleOneTile,
    %   fuseTile, multiBandBlending / linear / none, void paint, uint8) is ONE device call, aps_mex('render'); the crop
    %   (:430-432, cropNonzeroBbox) is a device reduction.  Gain compensation still runs the reference's own
    %   gainCompensationRKf (host solve) when opts.gainCompensation is set.
    %   Deliberate differences: opts.tile defaults to [2048 2048] clamped to the canvas (the reference derives it from free
    %   memory, :269-298, which makes its multiband output machine dependent).
    %   rgbAnnotation (the panorama with every image's warped outline and number drawn in, :438-477 and :653-679, which
    %   displayPanorama.m:126-136 stores) exists only when opts.showPanoramaImgsNums && opts.showCropBoundingBox: such a call
    %   is a debugging display and is forwarded whole to the reference's own file (aps_call_shadowed), whose local
    %   allWarpedBoxes / insertShape / insertText code draws it - like the display switches of the other shadows.
    if nargin < 7, opts = struct(); end
    if isfield(opts, 'showPanoramaImgsNums') && isfield(opts, 'showCropBoundingBox') && ...
            all(logical(opts.showPanoramaImgsNums)) && all(logical(opts.showCropBoundingBox))
        [panorama, rgbAnnotation] = aps_call_shadowed('renderPanorama', mfilename('fullpath'), input, images, imgSize, cameras, mode, refIdx, opts);
        return
    end
    opts = fillDefaults(opts, cameras, refIdx);
    numImages = numel(images);
    rgbAnnotation = [];   % (:75, the reference's value when the two switches are not both set)
    if cameras(1).noRotation == 1 || (isfield(input, 'forcePlanarScan') && input.forcePlanarScan)
        panorama = planarScanPanorama(images, cameras, numImages, opts);   % :519-699 over the imageWarp shadow
        return
    end
    mode = lower(char(mode));
    planeMode = any(strcmp(mode, {'planar', 'perspective', 'stereographic'}));
    if planeMode && opts.autoRef                                            % :84-122
        best = inf;
        for ii = 1:numImages
            b = surfaceBounds(mode, cameras, imgSize, cameras(ii).R, opts);
            if strcmp(mode, 'stereographic')
                ext = max(abs(b)) * (1 + 2 * opts.margin) + opts.pixelPad / opts.fPan;
                side = max(1, ceil(2 * opts.fPan * ext * opts.resScale));
                area = double(side) * double(side);
            else
                d = [b(2) - b(1), b(4) - b(3)];
                lo = [b(1) b(3)] - opts.margin * d - opts.pixelPad / opts.fPan;
                hi = [b(2) b(4)] + opts.margin * d + opts.pixelPad / opts.fPan;
                area = double(max(1, ceil(opts.fPan * (hi(1) - lo(1)) * opts.resScale))) * ...
                       double(max(1, ceil(opts.fPan * (hi(2) - lo(2)) * opts.resScale)));
            end
            if area < best, best = area; refIdx = ii; end
        end
    end
    b = surfaceBounds(mode, cameras, imgSize, cameras(refIdx).R, opts);     % [aMin aMax bMin bMax]
    if strcmp(mode, 'stereographic')
        ext = max(abs(b)); b = [-ext ext -ext ext];                        % centred square, :196-201
    end
    d = [b(2) - b(1), b(4) - b(3)];
    b = b + opts.margin * [-d(1) d(1) -d(2) d(2)];
    if planeMode, b = b + (opts.pixelPad / opts.fPan) * [-1 1 -1 1]; end
    W = max(1, ceil(opts.fPan * (b(2) - b(1)) * opts.resScale));
    H = max(1, ceil(opts.fPan * (b(4) - b(3)) * opts.resScale));
    if planeMode                                                            % global pixel cap, :170-177
        maxPixel = round(opts.maxMegapixel * 1e6);
        if double(H) * double(W) > maxPixel
            opts.resScale = opts.resScale * sqrt(maxPixel / (double(H) * double(W)));
            W = max(1, ceil(opts.fPan * (b(2) - b(1)) * opts.resScale));
            H = max(1, ceil(opts.fPan * (b(4) - b(3)) * opts.resScale));
        end
    end
    o0 = b(1); o1 = b(3);
    if isempty(opts.tile)
        side = min([H, W]);
        if side >= 512, side = max(512, min([2048, H, W])); end
        opts.tile = [side side];
    end
    gains = ones(numImages, 3);
    if opts.gainCompensation                                                % :303-330, the reference's own function
        srcW = cell(1, numImages);
        switch mode
            case {'planar', 'perspective', 'stereographic'}
                gains = gainCompensationRKf(images, cameras, mode, refIdx, opts, H, W, o0, o1, [], [], [], srcW);
            case 'cylindrical'
                gains = gainCompensationRKf(images, cameras, mode, refIdx, opts, H, W, [], [], o0, o1, [], srcW);
            otherwise
                gains = gainCompensationRKf(images, cameras, mode, refIdx, opts, H, W, [], [], o0, [], o1, srcW);
        end
    end
    panorama = aps_renderTiles(images, cameras, mode, refIdx, opts, H, W, o0, o1, double(gains));
    if opts.cropBorder
        rect = aps_mex('crop_nonzero_bbox', panorama, double(strcmpi(opts.canvasColor, 'white')));
        panorama = panorama(rect(1):rect(2), rect(3):rect(4), :);
    end
end

function opts = fillDefaults(opts, cameras, refIdx)
    d = {'fPan', cameras(refIdx).K(1, 1); 'resScale', 1.0; 'anglePower', 1; 'cropBorder', true; 'margin', 0.01; ...
         'tile', []; 'maxMegapixel', 50; 'robustPct', [1 99]; 'uvAbsCap', 8.0; 'pixelPad', 24; 'autoRef', true; ...
         'canvasColor', 'black'; 'gainCompensation', true; 'sigmaN', 10.0; 'sigmag', 0.1; 'blending', 'multiband'; ...
         'overlapStride', 4; 'pyrLevels', 3; 'pyrSigma', 1.0; 'composeNonePolicy', 'last'};
    for k = 1:size(d, 1)
        if ~isfield(opts, d{k, 1}), opts.(d{k, 1}) = d{k, 2}; end
    end
end

function rays = sampleRays(cam, H, W, border)
    % 48 x 32 interior grid (+ 4 x border edge samples): pixel -> K \ [u v 1]' -> R' (world rays, 3 x M)
    [U, V] = meshgrid(linspace(1, W, 48), linspace(1, H, 32));
    u = U(:)'; v = V(:)';
    if border > 0
        xb = linspace(1, W, border); yb = linspace(1, H, border);
        u = [u, xb, xb, ones(1, border), W * ones(1, border)];
        v = [v, ones(1, border), H * ones(1, border), yb, yb];
    end
    rays = cam.R' * (cam.K \ [u; v; ones(1, numel(u))]);
end

function b = surfaceBounds(mode, cams, imgSize, Rref, opts)
    % [aMin aMax bMin bMax] of cylindricalBounds / sphericalBounds / planarBounds / stereographicBounds (:1507-1754)
    b = [inf -inf inf -inf];
    robust = any(strcmp(mode, {'planar', 'perspective', 'stereographic'}));
    for i = 1:numel(cams)
        r = sampleRays(cams(i), imgSize(i, 1), imgSize(i, 2), 512 * robust);
        switch mode
            case 'cylindrical'
                p = atan2(r(1, :), r(3, :)); q = r(2, :) ./ hypot(r(1, :), r(3, :));
            case {'spherical', 'equirectangular'}
                p = atan2(r(1, :), r(3, :)); q = atan2(r(2, :), hypot(r(1, :), r(3, :)));
            case {'planar', 'perspective'}
                rr = Rref * r; keep = rr(3, :) > 1e-4;
                if ~any(keep), continue, end
                p = rr(1, keep) ./ rr(3, keep); q = rr(2, keep) ./ rr(3, keep);
            otherwise % stereographic
                rr = Rref * r; rr = rr ./ sqrt(sum(rr .^ 2, 1));
                den = 1 + rr(3, :); keep = den > 1e-6;
                if ~any(keep), continue, end
                p = rr(1, keep) ./ den(keep); q = rr(2, keep) ./ den(keep);
        end
        if robust
            if isfinite(opts.uvAbsCap) && opts.uvAbsCap > 0
                p = max(-opts.uvAbsCap, min(opts.uvAbsCap, p)); q = max(-opts.uvAbsCap, min(opts.uvAbsCap, q));
            end
            pr = prctile(p, opts.robustPct); qr = prctile(q, opts.robustPct);
            b = [min(b(1), pr(1)), max(b(2), pr(2)), min(b(3), qr(1)), max(b(4), qr(2))];
        else
            b = [min(b(1), min(p)), max(b(2), max(p)), min(b(3), min(q)), max(b(4), max(q))];
        end
    end
    if robust
        if ~all(isfinite(b(1:2))) || b(1) >= b(2), b(1:2) = [-1 1]; end
        if ~all(isfinite(b(3:4))) || b(3) >= b(4), b(3:4) = [-1 1]; end
    end
end

function panorama = planarScanPanorama(images, cameras, numImages, opts)
    % pureNonRotationalPanoramas (:519-699): canvas = bounding box of the H2refined corner maps, every image and its
    % tent weight warped onto it (imageWarp shadow -> device), whole-canvas 'none' / 'linear' / 'multiband'.
    lims = zeros(numImages, 4);
    for k = 1:numImages
        [xl, yl] = outputLimitsScratch(cameras(k).H2refined, [1 size(images{k}, 2)], [1 size(images{k}, 1)]);
        lims(k, :) = [xl yl];
    end
    xMin = min(lims(:, 1)); xMax = max(lims(:, 2)); yMin = min(lims(:, 3)); yMax = max(lims(:, 4));
    width = round(xMax - xMin); height = round(yMax - yMin);
    view = imref2dScratch([height width], [xMin xMax], [yMin yMax]);
    Iw = cell(1, numImages); Ww = cell(1, numImages);
    for k = 1:numImages
        [h, w, ~] = size(images{k});
        wx = ones(1, w); wx(1:ceil(w / 2)) = linspace(0, 1, ceil(w / 2)); wx(floor(w / 2) + 1:w) = linspace(1, 0, w - floor(w / 2));
        wy = ones(1, h); wy(1:ceil(h / 2)) = linspace(0, 1, ceil(h / 2)); wy(floor(h / 2) + 1:h) = linspace(1, 0, h - floor(h / 2));
        Iw{k} = imageWarp(single(images{k}) / 255, cameras(k).H2refined, view);
        Ww{k} = max(0, min(1, imageWarp(single(wy' * wx), cameras(k).H2refined, view)));
    end
    switch lower(opts.blending)
        case 'none'
            [~, idx] = max(cat(3, Ww{:}), [], 3);
            panorama = zeros(height, width, 3, 'single');
            for k = 1:numImages
                m = repmat(idx == k, [1 1 3]); panorama(m) = Iw{k}(m);
            end
        case 'linear'
            panorama = linearBlending(Iw, Ww);
        otherwise
            panorama = multiBandBlending(Iw, Ww, opts.pyrLevels, true, opts.pyrSigma);
    end
    void = ~any(cat(3, Ww{:}) > 0, 3);
    panorama(repmat(void, [1 1 3])) = double(strcmpi(opts.canvasColor, 'white'));
    panorama = uint8(max(0, min(255, round(255 * panorama))));
end
