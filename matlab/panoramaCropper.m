function croppedImage = panoramaCropper(input, stitchedImage)
    %PANORAMACROPPER Shadows PP/imageProcessing/panoramaCropper.m: same fields, same rectangle, same fallback.
    %   The gray/threshold/imfill pass and the line-by-line rectangle scan (:73-157 of the reference) run on the device
    %   through aps_mex('crop_rect'); the crop itself, the warning and the optional bounding-box figure stay here.
    need = ["canvasColor", "blackRange", "whiteRange", "showCropBoundingBox", "displayPanoramas"];
    absent = need(~isfield(input, need));
    if ~isempty(absent)
        error('panoramaCropper:MissingField', 'Missing required input fields: %s', strjoin(absent, ', '));
    end
    colour = lower(string(input.canvasColor));
    if ~(colour == "black" || colour == "white")
        error('panoramaCropper:InvalidCanvasColor', 'input.canvasColor must be "black" or "white".');
    end
    okRange = @(v) isscalar(v) && isnumeric(v) && isfinite(v) && v >= 0 && v <= 255;
    if ~okRange(input.blackRange)
        error('panoramaCropper:InvalidBlackRange', 'input.blackRange must be a numeric scalar in [0,255].');
    end
    if ~okRange(input.whiteRange)
        error('panoramaCropper:InvalidWhiteRange', 'input.whiteRange must be a numeric scalar in [0,255].');
    end
    if ~(islogical(input.showCropBoundingBox) && isscalar(input.showCropBoundingBox)) || ...
            ~(islogical(input.displayPanoramas) && isscalar(input.displayPanoramas))
        error('panoramaCropper:InvalidFlag', 'input.showCropBoundingBox and input.displayPanoramas must be logical scalars.');
    end

    white = colour == "white";
    if white, level = double(input.whiteRange); else, level = double(input.blackRange); end
    img = stitchedImage;
    if ~isa(img, 'uint8'), img = uint8(img); end  % the reference's thresholds are 8-bit too
    [rect, inside] = aps_mex('crop_rect', img, white, level);
    offsetx = rect(1); offsety = rect(2); cropW = rect(3); cropH = rect(4);

    if inside
        croppedImage = stitchedImage(offsety:offsety + cropH, offsetx:offsetx + cropW, :);
    else
        warning('Cannot crop the image. Image has background holes.');
        croppedImage = stitchedImage;
    end

    if input.showCropBoundingBox && input.displayPanoramas
        figure('Name', 'Crop rectangle panorama image');
        imshow(stitchedImage);
        hold on
        rectangle('Position', [offsetx offsety cropW cropH], 'EdgeColor', 'r', 'LineWidth', 2, 'LineStyle', '--')
        hold off
        exportgraphics(gcf, 'pano_bbox.jpg');
    end
end
